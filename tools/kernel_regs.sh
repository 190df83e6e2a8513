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
  # one line per kernel; a kernel's block in the notes starts at "- .agpr_count" (its LDS size comes BEFORE its name)
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | python3 -c '
import sys
cur = {}
def flush():
    if cur.get("name"):
        print("%-72s vgpr %3s sgpr %3s spills %3s scratch %5s B  lds %6s B" % (cur["name"][:72], cur.get("vgpr_count", "?"), cur.get("sgpr_count", "?"),
              cur.get("vgpr_spill_count", "?"), cur.get("private_segment_fixed_size", "?"), cur.get("group_segment_fixed_size", "?")))
for line in sys.stdin:
    t = line.strip()
    if t.startswith("- .agpr_count"):
        flush(); cur = {}
        continue
    for k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        if t.startswith("." + k + ":"):
            cur[k] = t.split(":", 1)[1].strip()
flush()'
done
rm -rf $T
