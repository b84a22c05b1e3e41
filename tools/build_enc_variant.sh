#!/bin/bash
# A/B experiments on the encoder kernels: builds zeekstd_amd/libzk_<tag>.so from extra -D flags (only zk_encode.hip is recompiled)
#   tools/build_enc_variant.sh clk:-DZKE_CLOCKS
set -e
cd "$(dirname "$0")/../zeekstd_amd/csrc"
make -j8 >/dev/null
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c zk_encode.hip -o build/var_$tag.o
  objs=$(ls build/*.o | grep -v "zk_encode" | grep -v "/var_" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,--as-needed -pthread -ldl -o ../libzk_$tag.so build/var_$tag.o $objs
done
