# usage: bash tools/build_alt_ref.sh <name> <git ref>   -> gsgen_amd/lib_alt/<name>.so
# The whole library as it was at <git ref> (every .hip file from that tree), for same-box A/Bs against the working tree through
# GSGEN_HIP_LIB (tools/ab.sh): e.g. `bash tools/build_alt_ref.sh pre HEAD` before measuring an uncommitted change.
set -e
name=$1; ref=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/alt_ref_$name; rm -rf $T; mkdir -p $T $R/gsgen_amd/lib_alt
(cd $R && git archive $ref gsgen_amd/csrc include) | tar -x -C $T
cd $T
for f in composite composite_bwd geometry binning legacy; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Igsgen_amd/csrc -Iinclude -c gsgen_amd/csrc/$f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gsgen_amd/lib_alt/$name.so composite.o composite_bwd.o geometry.o binning.o legacy.o
ls -la $R/gsgen_amd/lib_alt/$name.so
