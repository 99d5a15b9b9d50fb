#!/bin/bash
# Builds m3p_amd/libm3p_hip_<name>.so: <file>.hip recompiled with extra -D flags, every other object taken from the
# production build in m3p_amd/csrc (run make first).   usage: tools/build_variant.sh <name> <file.hip> -DFOO=1 ...
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../m3p_amd/csrc"
mkdir -p /tmp/m3p_var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics \
  -Wno-unused-result -ffp-contract=fast "$@" -c $file -o /tmp/m3p_var_$name/${file%.hip}.o 2>/dev/null
objs=""
for f in *.hip; do
  if [ "$f" == "$file" ]; then objs="$objs /tmp/m3p_var_$name/${f%.hip}.o"; else objs="$objs ${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libm3p_hip_$name.so $objs
echo built libm3p_hip_$name.so
