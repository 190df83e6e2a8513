#!/bin/bash
# VGPR / scratch / LDS per kernel of the built library (reads the code-object notes)
LIB=${1:-basevar_amd/lib/libbasevar_amd.so}
T=$(mktemp -d)
python3 - "$LIB" "$T" <<'PY'
import sys
data=open(sys.argv[1],'rb').read()
# the fat binary embeds gfx950 ELF code objects: carve every ELF whose e_machine is AMDGPU (224)
import struct
n=0; pos=0
while True:
    pos=data.find(b'\x7fELF',pos)
    if pos<0: break
    if struct.unpack_from('<H',data,pos+18)[0]==224:
        shoff=struct.unpack_from('<Q',data,pos+40)[0]; shentsize,shnum=struct.unpack_from('<HH',data,pos+58)
        open('%s/co%d.elf'%(sys.argv[2],n),'wb').write(data[pos:pos+shoff+shentsize*shnum]); n+=1
    pos+=4
PY
for f in $T/*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count" | sed 's/^ *//' | paste -sd' ' | sed 's/\.name:/\n.name:/g'
done
echo
rm -rf $T
