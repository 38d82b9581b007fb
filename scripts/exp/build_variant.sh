#!/bin/bash
# scripts/exp/build_variant.sh <name> <file without .hip> "<extra flags>": a build of the library with ONE translation unit compiled
# differently -> scripts/exp/variants/<name>/libposelib_amd.so (A/B timing on one box: POSELIB_AMD_LIB=...)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R/poselib_amd/csrc
make -s
mkdir -p $R/scripts/exp/variants/$1
X=""
[ $2 = kernels ] && X="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $X $3 -c $2.hip -o /tmp/variant_$1_$2.o
OBJS=""
for o in kernels gen_rel lm_cam focal sfocal pipeline driver; do
  if [ $o = $2 ]; then OBJS="$OBJS /tmp/variant_$1_$2.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $R/scripts/exp/variants/$1/libposelib_amd.so $OBJS
echo built $1
