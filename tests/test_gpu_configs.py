"""BASELINE.json configurations on the GPU: parity against the oracle at sizes the oracle finishes
in seconds, and size-independent properties (shard additivity, symmetry, bounds, known coupling
structure) at the full sizes."""
import time

import numpy as np
import pytest

from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
FS = 1000.0


def synth(T, R, C, tone, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, R, C)).astype(np.float32)
    t = np.arange(T) / FS
    x += (0.5 * np.sin(2 * np.pi * tone * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)).astype(np.float32)
    return x


def close(a, b, rtol, atol_scale, what):
    ok = ~np.isnan(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    scale = np.abs(b[ok]).max()
    err = np.abs(a[ok] - b[ok])
    worst = (err / (rtol * np.abs(b[ok]) + atol_scale * scale)).max()
    assert worst <= 1.0, f"{what}: max err {err.max():.3e}, scale {scale:.3e}, err/bound {worst:.2f}"


@pytest.fixture(scope="module")
def sc():
    import spectral_connectivity_amd as pkg
    return pkg


def test_cfg2_full_size_csm_coherency(sc):
    """configs[1]: 32 ch x 100 trials x 1024 samples, 5 tapers, single window."""
    x = synth(1024, 100, 32, 40.0, 2)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=FS, NW=3)
    csm = so.expectation_csm_gemm(coef)
    close(c._expectation_cross_spectral_matrix(), csm[:, :513], 1e-5, 1e-5, "csm")
    close(c.coherency(), so.coherency(coef, csm=csm), 1e-5, 1e-5, "coherency")
    close(c.coherence_magnitude(), so.coherence_magnitude(coef, csm=csm), 1e-5, 1e-5, "coherence")


def _wpli_chunked(coef):
    """wPLI of the oracle without the (W,R,K,N,C,C) temporary: loop over (window, trial)."""
    W, R, K, N, C = coef.shape
    F = N // 2 + 1
    s_im = np.zeros((W, F, C, C))
    s_abs = np.zeros((W, F, C, C))
    for w in range(W):
        for r in range(R):
            X = coef[w, r, :, :F, :]
            im = (X[..., :, None] * X[..., None, :].conj()).imag
            s_im[w] += im.sum(axis=0)
            s_abs[w] += np.abs(im).sum(axis=0)
    n = R * K
    idx = np.arange(C)
    s_im[..., idx, idx] = 0
    s_abs[..., idx, idx] = 0
    wgt = s_abs / n
    wgt[wgt < so.EPS] = 1
    return (s_im / n) / wgt


def test_cfg3_reduced_trials_coherence_wpli(sc):
    """configs[2] geometry (128 ch, NW=4, 256-pt windows, step 128) with 6 trials vs the oracle."""
    x = synth(1024, 6, 128, 60.0, 3)
    kw = dict(n_time_samples_per_window=256, n_time_samples_per_step=128)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=4, **kw)
    c = sc.Connectivity.from_multitaper(m)
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=FS, NW=4, **kw)
    csm = so.expectation_csm_gemm(coef)
    close(c.coherence_magnitude(), so.coherence_magnitude(coef, csm=csm), 1e-5, 1e-5, "coherence")
    close(c.weighted_phase_lag_index(), _wpli_chunked(coef), 1e-5, 1e-5, "wpli")


def test_cfg3_full_size_properties(sc):
    """configs[2] at full size (1000 trials): properties that need no reference run."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    x = synth(1024, 1000, 128, 60.0, 3)
    kw = dict(n_time_samples_per_window=256, n_time_samples_per_step=128)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=4, **kw)
    c = sc.Connectivity.from_multitaper(m)
    coh, wpli = c.coherence_magnitude(), c.weighted_phase_lag_index()
    assert coh.shape == wpli.shape == (7, 129, 128, 128)
    off = ~np.eye(128, dtype=bool)
    assert np.isnan(coh[..., ~off]).all() and np.all((coh[..., off] >= 0) & (coh[..., off] <= 1))
    assert np.all(np.abs(wpli) <= 1 + 1e-6) and np.all(wpli[..., ~off] == 0)
    np.testing.assert_allclose(coh, np.swapaxes(coh, -1, -2), rtol=0, atol=0, equal_nan=True)     # Hermitian mirror
    np.testing.assert_allclose(wpli, -np.swapaxes(wpli, -1, -2), rtol=0, atol=0)                   # antisymmetric
    f60 = int(round(60.0 / (FS / 256)))
    assert np.nanmean(coh[:, f60]) > 5 * np.nanmean(coh[:, 100])        # shared 60 Hz tone, white elsewhere
    # shard additivity (the multi-GPU sum rule): accumulators of trials [0,400) + [400,1000) == all trials
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    h = torch.from_numpy(np.ascontiguousarray(m.tapers.T / FS, dtype=np.float32)).cuda()
    def acc(lo, hi):
        xs = torch.from_numpy(x[:, lo:hi]).cuda()
        sp = engine.multitaper_spectra(xs, h, 256, 128, 256, 7, "constant")
        return engine.accumulate(sp, "trials_tapers", planes)[0]
    whole, parts = acc(0, 1000), acc(0, 400) + acc(400, 1000)
    rel = (whole - parts).abs().max() / whole.abs().max()
    assert rel < 2e-5, rel      # two different f32 summation orders over 7000 observations
    # stage B at FULL n_obs against an fp64 contraction of the same device spectra (torch, GPU):
    # bounds the f32 / bf16x3 accumulation error where the reference itself cannot run
    sp = m.device_spectra()
    X = sp.X.reshape(129, 7, 7000, 128)
    csm_ref = torch.empty((7, 129, 128, 128), dtype=torch.complex128, device="cuda")
    for f0 in range(0, 129, 8):
        Xd = X[f0:f0 + 8].to(torch.complex128)                          # (f, w, o, c)
        csm_ref[:, f0:f0 + 8] = torch.einsum("fwoc,fwod->wfcd", Xd, Xd.conj()) / 7000.0
        del Xd
    csm_ref = csm_ref.cpu().numpy()
    csm = c._expectation_cross_spectral_matrix()
    power_ref = np.real(np.einsum("wfcc->wfc", csm_ref))
    err_p = np.abs(c.power() - power_ref).max() / power_ref.max()
    err_s = np.abs(csm - csm_ref).max() / np.abs(csm_ref).max()
    norm = np.sqrt(power_ref[..., :, None] * power_ref[..., None, :])
    coh_ref = np.abs(csm_ref / norm) ** 2
    d = np.abs(coh - coh_ref)[..., off]
    bound = 1e-5 * coh_ref[..., off] + 1e-5 * np.nanmax(coh_ref[..., off])
    print(f"full cfg3 stage-B error vs fp64: power {err_p:.2e}, csm {err_s:.2e}, coherence worst err/bound {np.max(d / bound):.2f}")
    assert err_p < 3e-6 and err_s < 3e-6 and np.max(d / bound) <= 0.5


def test_cfg4_granger_reduced_vs_oracle_and_full_size(sc):
    """configs[3]: 64 ch x 200 trials x 4096 samples, all 2016 pairs through the batched Wilson kernel."""
    rng = np.random.default_rng(4)
    C, T = 64, 4096
    # sparse stable VAR(2): channel c is driven by c-1 (lag 1) and c-5 (lag 2)
    def simulate(R):
        e = rng.standard_normal((T + 200, R, C))
        y = np.zeros_like(e)
        for t in range(2, T + 200):
            y[t] = 0.5 * y[t - 1] - 0.3 * y[t - 2] + e[t]
            y[t, :, 1:] += 0.35 * y[t - 1, :, :-1]
            y[t, :, 5:] += 0.25 * y[t - 2, :, :-5]
        return y[200:]
    x = simulate(12)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    pairs = [(0, 1), (3, 8), (10, 40), (62, 63)]
    got = c.subset_pairwise_spectral_granger_prediction(pairs)
    coef, _ = so.multitaper_fft(x, fs=FS, NW=3)
    ref = so.pairwise_spectral_granger_prediction(coef, pairs=pairs)
    from conftest import granger_close
    granger_close(got, ref, 3e-5, what="cfg4 subset")
    # full size: 200 trials, every pair
    x = simulate(200).astype(np.float32)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    t0 = time.perf_counter()
    gp = c.pairwise_spectral_granger_prediction()
    dt = time.perf_counter() - t0
    print(f"cfg4 full: 2016 pairs in {dt:.2f} s, Wilson iterations {c._last_wilson['iterations']}, "
          f"not converged {c._last_wilson['not_converged']}")
    assert gp.shape == (1, 2049, 64, 64) and c._last_wilson["not_converged"] == 0
    band = slice(20, 900)
    drive = np.nanmean(np.nan_to_num(gp[0, band, np.arange(1, 64), np.arange(0, 63)]))      # c-1 -> c
    back = np.nanmean(np.nan_to_num(gp[0, band, np.arange(0, 63), np.arange(1, 64)]))       # c -> c-1
    far = np.nanmean(np.nan_to_num(gp[0, band, 0, 30:60]))
    assert drive > 10 * back and drive > 10 * far, (drive, back, far)


def test_cfg5_canonical_coherence_reduced_vs_oracle_and_full_size(sc):
    """configs[4]: 256 ch, 16 groups of 16; oracle SVD form at 24 trials, properties at 500."""
    rng = np.random.default_rng(5)
    C, T = 256, 1024
    labels = np.repeat(np.arange(16), 16)
    t = np.arange(T) / FS
    def make(R):
        x = rng.standard_normal((T, R, C)).astype(np.float32)
        src = rng.standard_normal((T, R, 16)).astype(np.float32)
        x += 0.6 * np.repeat(src, 16, axis=2)                  # each group shares a latent source
        x[:, :, :32] += (0.8 * np.sin(2 * np.pi * 30 * t)[:, None, None] * rng.standard_normal((1, R, 1))).astype(np.float32)
        return x
    x = make(24)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    cc, lab = c.canonical_coherence(labels)
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=FS, NW=3)
    ref, _ = so.canonical_coherence(coef[..., :129, :][:, :, :, :, :] if False else coef, labels)
    close(cc, ref, 5e-5, 5e-5, "canonical coherence (256 ch, 24 trials)")
    x = make(500)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    t0 = time.perf_counter()
    cc, lab = c.canonical_coherence(labels)
    print(f"cfg5 full: canonical coherence in {time.perf_counter() - t0:.2f} s")
    assert cc.shape == (1, 513, 16, 16)
    off = ~np.eye(16, dtype=bool)
    assert np.all((cc[..., off] >= 0) & (cc[..., off] <= 1 + 1e-9)) and np.isnan(cc[..., ~off]).all()
    np.testing.assert_array_equal(cc, np.swapaxes(cc, -1, -2))
    f30 = int(round(30.0 / (FS / 1024)))
    assert cc[0, f30, 0, 1] > 0.5 and cc[0, f30, 0, 1] > 3 * np.nanmedian(cc[0, f30][2:, 2:][~np.eye(14, dtype=bool)])


@pytest.mark.parametrize("world", [2, 3])
def test_trial_sharded_pipeline_ranks_share_one_gpu(world):
    """parallel.sharded_measures (accumulate -> reduce-scatter -> epilogue -> gather, pipelined over frequency
    groups on a second stream) and parallel.ShardedConnectivity (measures, Granger pairs dealt out over the ranks,
    canonical-coherence bins split) with 2 / 3 ranks sharing this GPU over gloo equal the single-process results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29541 + world),
                          os.path.join(root, "tools", "check_sharded.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sharded_measures OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert "ShardedConnectivity OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_eight_rank_rehearsal_at_the_headline_size():
    """What the 8-GPU run of the bench does, rehearsed on the one GPU a test box has: 8 ranks (gloo, sharing the device) with
    125 of the 1000 trials of the cfg3 shape each -- planes-format stage A and B per rank, direct exchange of the bin blocks,
    the epilogue kernel summing the eight received blocks in rank order, gather on rank 0 -- against the FLOAT64 reference over
    all 1000 trials (tests/fp64_device_ref.py, the bound of tests/test_gpu_full_depth.py) and against the single-process result."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SC_SHARD_FULL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", "29557",
                          os.path.join(root, "tools", "check_sharded.py")],
                         env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "sharded_measures OK (full size, 8 ranks x 125 trials" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("algorithm", ["direct", "ring", "library"])
def test_rccl_exchange_path_world_one(algorithm):
    """The RCCL calls of the N > 1 path (all_to_all_single of the direct exchange / reduce_scatter_tensor of the ring one /
    sc_comm_exchange_blocks_f32 of the C ABI's own communicator,
    gather, all_gather_into_tensor, all_reduce on the `nccl` backend, exchange stream, padded buffers) rehearsed with the one
    GPU a test box has: a one-rank nccl group, SC_FORCE_EXCHANGE=1 so that nothing short-cuts the collectives."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SC_BENCH_BACKEND="nccl", SC_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", SC_EXCHANGE=algorithm)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29547",
                          os.path.join(root, "tools", "check_sharded.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sharded_measures OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert "ShardedConnectivity OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses) starts its own ranks and prints
    ONE JSON line from rank 0; on this 1-GPU box the ranks share the device over gloo (said in the line)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SC_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--trials", "64"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["exchange"] is not None and rec["value"] > 0


def test_hot_kernels_are_bit_reproducible():
    """Stage A and stage B (split bins, L2-atomic folds, direct HBM->LDS row loads, one LDS-only barrier per chunk)
    repeat bit-exactly: every sum has one writer and a fixed order, and a race would show up here."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    torch.manual_seed(0)
    x = torch.randn(1024, 200, 128, device="cuda")
    tap = torch.randn(7, 256, device="cuda")
    sp = engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant")
    x0 = sp.X.clone()
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    a0 = engine.accumulate(sp, "trials_tapers", planes)[0].clone()
    for _ in range(40):
        sp2 = engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant")
        assert torch.equal(sp2.X, x0)
        assert torch.equal(engine.accumulate(sp2, "trials_tapers", planes)[0], a0)


@pytest.mark.parametrize("C,R", [(4, 12000), (16, 3000), (47, 1000)])
def test_small_channel_kernel_at_full_observation_counts(sc, C, R):
    """The f32 VALU stage-B kernel (<= 48 channels, odd counts through the zero pad channel) at observation counts of
    the BASELINE scale -- up to 84 000 rows per bin, split over slices and workgroups -- against fp64 contractions of the
    same device spectra (torch, GPU): CSM, sum |Im s|, sum (Im s)^2 and sum sign(Im s) per (window, bin, pair)."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(C)
    x = (rng.standard_normal((384, R, C)) + 0.6 * rng.standard_normal((384, R, 1))).astype(np.float32)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=4, n_time_samples_per_window=128,
                      n_time_samples_per_step=128)
    sp = m.device_spectra()
    W, K, F = 3, 7, 65
    X = sp.coefficients().to(torch.complex128).reshape(F, W, R * K, C)           # (f, w, o, c)
    n_obs = R * K
    csm = torch.zeros((F, W, C, C), dtype=torch.complex128, device="cuda")
    ab = torch.zeros((F, W, C, C), dtype=torch.float64, device="cuda")
    sq = torch.zeros_like(ab)
    sg = torch.zeros_like(ab)
    step = max(1, (1 << 22) // (C * C * W))
    for f0 in range(0, F, 4):
        for o0 in range(0, n_obs, step):
            Xc = X[f0:f0 + 4, :, o0:o0 + step]
            sc_ = Xc[..., :, None] * Xc[..., None, :].conj()                          # (f, w, o, c, d)
            csm[f0:f0 + 4] += sc_.sum(2)
            ab[f0:f0 + 4] += sc_.imag.abs().sum(2)
            sq[f0:f0 + 4] += (sc_.imag ** 2).sum(2)
            sg[f0:f0 + 4] += torch.sign(sc_.imag).sum(2)
            del sc_
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ
    a, n = engine.accumulate(sp, "trials_tapers", planes)
    assert n == n_obs
    got = engine.measure(a, C, planes, n, _lib.M_CSM).reshape(W, F, C, C).cpu().numpy()
    ref = (csm / n_obs).permute(1, 0, 2, 3).cpu().numpy()
    assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max()
    # wPLI = |sum Im| / sum |Im|, debiased = (|sum Im|^2 - sum Im^2) / ((sum |Im|)^2 - sum Im^2): exercises both planes
    off = ~np.eye(C, dtype=bool)
    im = csm.imag.permute(1, 0, 2, 3).cpu().numpy()
    abn, sqn = ab.permute(1, 0, 2, 3).cpu().numpy(), sq.permute(1, 0, 2, 3).cpu().numpy()
    wpli_ref = im / np.where(abn > 0, abn, 1.0)
    got = engine.measure(a, C, planes, n, _lib.M_WPLI).reshape(W, F, C, C).cpu().numpy()
    inner = slice(1, F - 1)                                   # DC / Nyquist: Im s = 0 exactly
    assert np.abs(got[:, inner][..., off] - wpli_ref[:, inner][..., off]).max() <= 1e-5
    with np.errstate(invalid="ignore", divide="ignore"):
        deb_ref = (im ** 2 - sqn) / (abn ** 2 - sqn)              # 0 / 0 on the (masked) diagonal
    got = engine.measure(a, C, planes, n, _lib.M_DEBIASED_WPLI2).reshape(W, F, C, C).cpu().numpy()
    assert np.abs(got[:, inner][..., off] - deb_ref[:, inner][..., off]).max() <= 1e-5
    a2, _ = engine.accumulate(sp, "trials_tapers", _lib.PLANE_SIGN_IM)
    got = engine.measure(a2, C, _lib.PLANE_SIGN_IM, n, _lib.M_PLI).reshape(W, F, C, C).cpu().numpy()
    pli_ref = (sg / n_obs).permute(1, 0, 2, 3).cpu().numpy()
    # sign() of an f32 product vs of an fp64 product: they differ only where |Im s| is within rounding of 0
    assert np.abs(np.abs(got[:, inner][..., off]) - np.abs(pli_ref[:, inner][..., off])).max() <= 8.0 / n_obs


def test_graphed_measures_replay_equals_the_eager_pass():
    """engine.GraphedMeasures: the three launches of a small request (configs[1]: 32 channels x 100 trials x 1024 samples, CSM +
    coherency) captured once in a hipGraph by the library; a replay on new data equals the eager pass bit for bit, and the
    oracle within the float32 engine's tolerance."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    from spectral_connectivity_amd.transforms import _make_tapers
    T, R, C, NW = 1024, 100, 32, 3
    tapers = _make_tapers(T, FS, NW, 5)
    h = torch.from_numpy(np.ascontiguousarray(tapers.T / FS, dtype=np.float32)).cuda()
    g = engine.GraphedMeasures((T, R, C), h, T, T, T, "constant", "trials_tapers", [_lib.M_COHERENCY, _lib.M_COHERENCE_MAGNITUDE])
    for seed in (2, 7):
        x = synth(T, R, C, 40.0, seed)
        coh_g, mag_g = [t.clone() for t in g(x)]
        coh_e, mag_e = g.eager()
        assert torch.equal(torch.view_as_real(coh_g).nan_to_num(), torch.view_as_real(coh_e).nan_to_num())
        assert torch.equal(mag_g.nan_to_num(), mag_e.nan_to_num())
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=FS, NW=NW)
    ref = so.coherency(coef)
    close(coh_g.reshape(ref.shape).cpu().numpy(), ref, 1e-5, 1e-5, "coherency from a graph replay")


def test_graphed_measures_owns_its_split_bin_workspace():
    """A captured pass replays the ADDRESS of stage B's split-bin scratch; the shared per-device scratch is replaced (and freed)
    by the next larger eager request.  GraphedMeasures keeps a scratch of its own: a replay after such a request still equals
    the eager pass bit for bit (few bins x many observations: every bin split over several workgroups)."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    from spectral_connectivity_amd.transforms import _make_tapers
    T, R, C, NW = 64, 3000, 32, 2
    tapers = _make_tapers(T, FS, NW, 3)
    h = torch.from_numpy(np.ascontiguousarray(tapers.T / FS, dtype=np.float32)).cuda()
    g = engine.GraphedMeasures((T, R, C), h, T, T, T, "constant", "trials_tapers", [_lib.M_COHERENCE_MAGNITUDE])
    assert g._ws, "this shape was chosen so that stage B splits its bins (a scratch buffer)"
    own = next(iter(g._ws.values()))
    shared = engine._ws_cache.get((own.device.type, own.device.index))
    assert shared is None or shared.data_ptr() != own.data_ptr()
    x = synth(T, R, C, 40.0, 3)
    first = g(x)[0].clone()
    # a larger eager request on the same device: the shared scratch grows (the old block goes back to the allocator) ...
    big = torch.randn((T, 4 * R, C), dtype=torch.float32, device="cuda")
    sp = engine.multitaper_spectra(big, h, T, T, T, 1, "constant")
    engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM)
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]       # ... and whatever was freed is overwritten
    again = g()[0]
    assert torch.equal(first.nan_to_num(), again.nan_to_num())
    assert torch.equal(again.nan_to_num(), g.eager()[0].nan_to_num())
    del junk


@pytest.mark.parametrize("dtype,C", [("float32", 6), ("float32", 7), ("float64", 6), ("float64", 5)])
def test_multitaper_takes_a_series_that_already_lives_in_hbm(sc, dtype, C):
    """Beyond the reference: ``Multitaper(time_series=<torch tensor on the GPU>)`` uses the tensor in place (no host round trip;
    bench.py's ``api`` pass).  Same numbers as the host-array path, bit for bit, in both engines; odd channel counts get their
    zero pad channel on the device; the NaN scan runs on the device with the first transform."""
    import warnings
    import torch
    rng = np.random.default_rng(11 + C)
    x = rng.standard_normal((512, 5, C)).astype(dtype)
    kw = dict(sampling_frequency=500.0, time_halfbandwidth_product=3, n_time_samples_per_window=128, n_time_samples_per_step=64)
    xd = torch.from_numpy(x).cuda()
    for cdtype in (np.complex64, np.complex128):
        host = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=cdtype)
        dev = sc.Connectivity.from_multitaper(sc.Multitaper(xd, **kw), dtype=cdtype)
        assert dev._shape5 == host._shape5 and dev.n_observations == host.n_observations
        for name in ("power", "coherence_magnitude", "weighted_phase_lag_index"):
            np.testing.assert_array_equal(getattr(dev, name)(), getattr(host, name)(), err_msg=f"{name} {np.dtype(cdtype)}")
    m = sc.Multitaper(xd, **kw)
    assert m.n_signals == C and m.n_trials == 5 and np.asarray(m.time_series).shape == x.shape
    np.testing.assert_array_equal(m.fft(), sc.Multitaper(x, **kw).fft())
    bad = xd.clone()
    bad[7, 1, 2] = float("nan")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sc.Multitaper(bad, **kw).fft()
    assert any("NaN or infinite" in str(i.message) for i in w), [str(i.message) for i in w]


@pytest.mark.parametrize("C", [258, 306, 401])
def test_more_than_256_signals(sc, C):
    """The reference has no channel limit (connectivity.py:447-526); one launch of the stage-B kernels stages at most 256 signals
    per observation row.  Beyond that engine.accumulate tiles the channels in blocks of 128 and assembles the record from the
    block pairs (engine._accumulate_blocked): a 306-channel MEG array -- or an odd 401 -- goes through every expectation measure
    of both engines, pairwise Granger on a subset, canonical coherence and (up to 512 signals) global coherence, against the oracle."""
    import spectral_connectivity_amd.options as options
    rng = np.random.default_rng(C)
    L, R = 64, 3
    x = rng.standard_normal((L, R, C)) + 0.6 * rng.standard_normal((L, R, 1))
    x[:, :, C - 3] += 0.8 * np.roll(x[:, :, 2], 3, axis=0)            # a lagged copy across the first and the last block
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2)
    F = L // 2 + 1
    csm = so.expectation_csm_gemm(coef)
    refs = {"power": so.power(coef), "coherence_magnitude": so.coherence_magnitude(coef, csm=csm)}
    if C <= 306:
        refs.update(weighted_phase_lag_index=so.weighted_phase_lag_index(coef), phase_locking_value=so.phase_locking_value(coef),
                    phase_lag_index=so.phase_lag_index(coef))
    old = options.precision
    try:
        for precision, dtype, tol in (("float32", np.complex64, 3e-5), ("float64", np.complex128, 1e-9)):
            options.precision = precision
            c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=dtype)
            for name, ref in refs.items():
                got = getattr(c, name)()
                ref = ref[:, :F]
                assert got.shape == ref.shape, (name, got.shape, ref.shape)
                assert np.array_equal(np.isnan(got), np.isnan(ref)), name
                ok = ~np.isnan(ref)
                err = np.abs(got[ok] - ref[ok]).max() / np.abs(ref[ok]).max()
                bound = 4.0 / c.n_observations if name == "phase_lag_index" else tol
                assert err <= bound, (precision, name, err)
            if C in (306, 401):
                # global coherence beyond 256 signals (round 5: the Householder / bisection kernel with a thread per row up to 512):
                # the leading squared singular values of the oracle's thin SVD, all two-sided bins
                vals, vecs = c.global_coherence(max_rank=2)
                ref_vals, _ = so.global_coherence(coef, max_rank=2)
                assert vals.shape == ref_vals.shape and vecs.shape[-2:] == (C, 2)
                gerr = np.abs(vals - ref_vals).max() / np.abs(ref_vals).max()
                print(f"  global coherence, {C} signals, {precision}: max err {gerr:.2e}")
                assert gerr <= (5e-5 if precision == "float32" else 1e-9), (precision, gerr)
            if C == 306:
                pairs = [(2, C - 3), (0, 130), (129, 300)]
                gp = c.subset_pairwise_spectral_granger_prediction(pairs)
                sub = sc.Connectivity.from_multitaper(sc.Multitaper(x[:, :, [2, C - 3]], **kw), dtype=dtype).pairwise_spectral_granger_prediction()
                a, b = gp[..., 2, C - 3], sub[..., 0, 1]
                both = ~np.isnan(a) & ~np.isnan(b)
                assert both.any() and np.abs(a[both] - b[both]).max() <= (2e-4 if precision == "float32" else 1e-8) * np.nanmax(b)
                labels = np.arange(C) // 4                                  # 77 groups of 4 channels, 9 observations per bin
                cc, _ = c.canonical_coherence(labels)
                ref_cc = np.asarray(so.canonical_coherence(coef, labels)[0])[:, :F]
                okc = ~np.isnan(ref_cc)
                assert cc.shape == ref_cc.shape and np.array_equal(np.isnan(cc), ~okc)
                err_cc = np.abs(cc[okc] - ref_cc[okc]).max()
                print(f"  canonical coherence, {C} signals, {precision}: max err {err_cc:.2e}")
                assert err_cc <= (5e-3 if precision == "float32" else 1e-6)
    finally:
        options.precision = old


@pytest.mark.parametrize("C", [258, 306, 307, 320, 418])
def test_more_than_256_signals_on_the_planes_format(sc, C, monkeypatch):
    """Round 6: planes-format spectra of MORE than 256 signals go straight to sc_fused2.hip, which plans its launches over any number of
    32-channel blocks (groups of four: triangles; pairs of blocks of different groups: 64 x 64 rectangles; an odd block count: the last
    block against every pair outside its group, the new launch shape (3, 2, 2)) -- no channel tiling on the host, no gathered copies.
    258 = 9 blocks (odd, the last one of 2 channels), 306 = 10, 307 = 10 with a zero pad channel, 320 = 10 whole blocks, 418 = 14 (the last group holds two blocks).
    Every accumulator family of the format against the oracle; the path is asserted (P is not None, no call of _accumulate_blocked)."""
    import spectral_connectivity_amd.engine as engine
    import spectral_connectivity_amd.options as options
    monkeypatch.setenv("SC_PLANES_MIN_CHANNELS", "44")                     # the format whatever the size of the request
    calls = []
    real = engine._accumulate_blocked
    monkeypatch.setattr(engine, "_accumulate_blocked", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    rng = np.random.default_rng(C)
    L, R = 64, 3
    x = rng.standard_normal((L, R, C)) + 0.6 * rng.standard_normal((L, R, 1))
    x[:, :, C - 3] += 0.8 * np.roll(x[:, :, 2], 3, axis=0)                  # a lagged copy across the first and the last block
    x[:, :, 140] += 0.7 * np.roll(x[:, :, 70], 2, axis=0)                   # ... and across two groups of four blocks
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2)
    F = L // 2 + 1
    refs = {"coherence_magnitude": so.coherence_magnitude(coef), "weighted_phase_lag_index": so.weighted_phase_lag_index(coef),
            "debiased_squared_weighted_phase_lag_index": so.debiased_squared_weighted_phase_lag_index(coef),
            "phase_lag_index": so.phase_lag_index(coef), "power": so.power(coef)}
    old = options.precision
    try:
        options.precision = "float32"
        for first in ("coherence_magnitude", "phase_lag_index", "debiased_squared_weighted_phase_lag_index"):
            c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=np.complex64)
            order = [first] + [n for n in refs if n != first]
            for name in order:
                got, ref = getattr(c, name)(), refs[name][:, :F]
                assert got.shape == ref.shape, (name, got.shape, ref.shape)
                assert np.array_equal(np.isnan(got), np.isnan(ref)), name
                ok = ~np.isnan(ref)
                err = np.abs(got[ok] - ref[ok]).max() / np.abs(ref[ok]).max()
                bound = 4.0 / c.n_observations if name == "phase_lag_index" else 3e-5
                assert err <= bound, (first, name, err)
            sp = c._device()
            assert sp.P is not None, "the spectra of this request should be in the planes format"
        assert not calls, "planes-format spectra beyond 256 signals must not be tiled on the host"
    finally:
        options.precision = old
