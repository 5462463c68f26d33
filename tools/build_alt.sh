# usage: bash tools/build_alt.sh <name> <patched composite.hip> [extra hipcc flags]   -> gsgen_amd/lib_alt/<name>.so
# An experiment build of the library: composite.hip replaced by the given file, every other object from the in-tree build
# (gsgen_amd/build/*.o, python -m gsgen_amd.build first).  For same-box A/B through GSGEN_HIP_LIB (tools/ab.sh).
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gsgen_amd/lib_alt /tmp/alt_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I$R/gsgen_amd/csrc "$@" -c $src -o /tmp/alt_$name/composite.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gsgen_amd/lib_alt/$name.so /tmp/alt_$name/composite.o $R/gsgen_amd/build/composite_bwd.o $R/gsgen_amd/build/geometry.o $R/gsgen_amd/build/binning.o $R/gsgen_amd/build/legacy.o
ls -la $R/gsgen_amd/lib_alt/$name.so
