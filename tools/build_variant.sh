#!/bin/bash
# usage: tools/build_variant.sh <tag> "<-D flags>" [file.hip ...]: zlib_amd/libzlib_amd_<tag>.so with the named sources (default
# zg_match.hip) compiled with the extra flags and the other objects taken from the regular build
set -eu
tag="$1"; flags="$2"; shift 2
srcs="${*:-zg_match.hip}"
cd "$(dirname "$0")/../zlib_amd/csrc"
make -j8 > /dev/null
objs=""
for o in zg_match zg_walk zg_entropy zg_misc zg_inflate zg_api zg_corpus; do
  if echo " $srcs " | grep -q " $o.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $flags -c $o.hip -o /tmp/${o}_$tag.o
    objs="$objs /tmp/${o}_$tag.o"
  else
    objs="$objs $o.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libzlib_amd_$tag.so $objs -pthread
echo built zlib_amd/libzlib_amd_$tag.so
