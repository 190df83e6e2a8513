#!/bin/bash
# code bytes per kernel of the built library
LIB=${1:-basevar_amd/lib/libbasevar_amd.so}
T=$(mktemp -d)
python3 - "$LIB" "$T" <<'PY'
import sys,struct
data=open(sys.argv[1],'rb').read()
n=0; pos=0
while True:
    pos=data.find(b'\x7fELF',pos)
    if pos<0: break
    if struct.unpack_from('<H',data,pos+18)[0]==224:
        shoff=struct.unpack_from('<Q',data,pos+40)[0]; shentsize,shnum=struct.unpack_from('<HH',data,pos+58)
        open('%s/co%d.elf'%(sys.argv[2],n),'wb').write(data[pos:pos+shoff+shentsize*shnum]); n+=1
    pos+=4
PY
for f in $T/*.elf; do /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide $f | awk '$4=="FUNC" {print $3, $8}'; done | sort -u -k2 | sort -n
rm -rf $T
