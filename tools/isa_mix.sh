#!/bin/bash
# Static instruction mix of the encoder kernels (scalar vs vector vs LDS vs scratch), for sizing
# VALU-issue pressure without a GPU. Usage: tools/isa_mix.sh [file.hip]
set -e
SRC=${1:-concentus_amd/csrc/celt_enc_kernel.hip}
TMP=$(mktemp -d)
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-gpu-rdc --cuda-device-only -S -o $TMP/k.s $SRC
awk '
/^_Z.*:$/ || /^[a-z_0-9]+kernel[a-z_0-9]*:$/ { name=$1 }
/^\t(s_|v_|ds_|scratch_|global_|buffer_|flat_)/ {
  split($1,a,"_"); cls=a[1]; if (cls=="scratch"||cls=="global"||cls=="buffer"||cls=="flat"||cls=="ds") ; c[name,cls]++; names[name]=1 }
/\.vgpr_count|\.sgpr_count|scratch_en|\.private_segment_fixed_size|\.group_segment_fixed_size/ { }
END { for (n in names) printf "%s s=%d v=%d ds=%d scratch=%d global=%d\n", n, c[n,"s"], c[n,"v"], c[n,"ds"], c[n,"scratch"], c[n,"global"] }' $TMP/k.s
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name):" $TMP/k.s | paste - - - - - | sed 's/\s\+/ /g'
rm -rf $TMP
