"""Minimal labelled arrays for the front end (wrapper.py) when the optional ``xarray`` package is absent.

``multitaper_connectivity`` / ``connectivity_to_xarray`` return ``xarray.DataArray`` / ``xarray.Dataset`` objects when
xarray can be imported (the reference's dependency, wrapper.py:8); without it they return the two classes below, which
carry the same information under the same attribute names -- ``values``, ``dims``, ``coords``, ``attrs``, ``name`` --
and the handful of operations the reference's tutorials use on the result (``sel`` by coordinate label, ``isel``,
``squeeze``, ``Dataset[name]`` / ``data_vars`` / iteration).  No arithmetic, no alignment: convert with
``xarray.DataArray(a.values, coords=a.coords, dims=a.dims, attrs=a.attrs, name=a.name)`` where more is needed.
"""
import numpy as np


class DataArray:
    def __init__(self, data, coords=None, dims=None, attrs=None, name=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
        if len(self.dims) != self.values.ndim:
            raise ValueError(f"{len(self.dims)} dimension names for a {self.values.ndim}-D array")
        if coords is None:
            coords = [np.arange(n) for n in self.values.shape]
        if isinstance(coords, dict):
            coords = [coords[d] for d in self.dims]
        self.coords = {}
        for d, c, n in zip(self.dims, coords, self.values.shape):
            c = np.asarray(c)
            if c.shape != (n,):
                raise ValueError(f"coordinate '{d}' has shape {c.shape}, the axis has length {n}")
            self.coords[d] = c
        self.attrs = dict(attrs or {})
        self.name = name

    shape = property(lambda self: self.values.shape)
    ndim = property(lambda self: self.values.ndim)
    dtype = property(lambda self: self.values.dtype)

    def __getitem__(self, key):
        """A coordinate by name (like xarray)."""
        return self.coords[key]

    def _take(self, dim, index):
        axis = self.dims.index(dim)
        values = np.take(self.values, index, axis=axis)
        if np.ndim(index) == 0:
            dims = self.dims[:axis] + self.dims[axis + 1:]
            coords = [self.coords[d] for d in dims]
        else:
            dims = self.dims
            coords = [self.coords[d][index] if d == dim else self.coords[d] for d in dims]
        return DataArray(values, coords=coords, dims=dims, attrs=self.attrs, name=self.name)

    def isel(self, **indexers):
        out = self
        for dim, index in indexers.items():
            out = out._take(dim, index)
        return out

    def sel(self, method=None, **indexers):
        """Select by coordinate label; ``method="nearest"`` picks the closest numeric label."""
        out = self
        for dim, label in indexers.items():
            coord = out.coords[dim]
            if method == "nearest":
                index = int(np.argmin(np.abs(coord.astype(float) - float(label))))
            else:
                hits = np.flatnonzero(coord == label)
                if hits.size == 0:
                    raise KeyError(f"{label!r} not found in coordinate '{dim}'")
                index = int(hits[0])
            out = out._take(dim, index)
        return out

    def squeeze(self):
        keep = [i for i, n in enumerate(self.values.shape) if n != 1]
        dims = [self.dims[i] for i in keep]
        return DataArray(self.values.squeeze(), coords=[self.coords[d] for d in dims], dims=dims, attrs=self.attrs,
                         name=self.name)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __repr__(self):
        dims = ", ".join(f"{d}: {n}" for d, n in zip(self.dims, self.values.shape))
        return f"<spectral_connectivity_amd DataArray {self.name!r} ({dims})>"


class Dataset(dict):
    """name -> DataArray (``xarray.Dataset`` stand-in: item access, ``data_vars``, iteration over names)."""

    @property
    def data_vars(self):
        return self

    @property
    def attrs(self):
        first = next(iter(self.values()), None)
        return dict(first.attrs) if first is not None else {}

    def __repr__(self):
        return f"<spectral_connectivity_amd Dataset with {list(self)}>"
