# usage: ab_lib_any.sh <tool.py> <lib.so> [<lib.so> ...]: the tool once per library, one process each
tool=$1; shift
for lib in "$@"; do echo "== $lib"; SC_HIP_LIB=$PWD/$lib python $tool 2>&1 | grep -v Warn | tail -3; done
