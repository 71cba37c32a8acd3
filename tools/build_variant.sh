#!/bin/bash
# an experimental build of libt1k_gpu.so for A/B runs: one device source compiled with extra flags, linked with the production objects
#   tools/build_variant.sh NAME "EXTRA FLAGS" t1k_chain.hip [more.hip ...]   -> t1k_amd/lib/variants/libt1k_NAME.so  (use with T1K_GPU_LIB=...)
set -e
N=$1; F=$2; shift 2
cd "$(dirname "$0")/../t1k_amd/csrc"
make -s -j8 all > /dev/null
mkdir -p build/var_$N ../lib/variants
OBJ=""
for o in build/*.o; do
  b=$(basename $o .o); repl=""
  for src in "$@"; do [ "$b" = "$(basename $src .hip)" ] && repl=$src; done
  if [ -n "$repl" ]; then /opt/rocm/bin/hipcc $F -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -c -o build/var_$N/$b.o $repl; OBJ="$OBJ build/var_$N/$b.o"; else OBJ="$OBJ $o"; fi
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o ../lib/variants/libt1k_$N.so $OBJ -lz -lpthread -ldl
echo "built t1k_amd/lib/variants/libt1k_$N.so"
