#!/bin/bash
# md5 of the disassembled text of every kernel in an object / library (to check that a refactor left codegen alone)
F=$1; T=$(mktemp -d)
python3 - "$F" "$T" <<'PY'
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
for f in $T/*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn --no-leading-addr $f | python3 -c "
import sys,hashlib,re
cur=None; h={}
for l in sys.stdin:
    m=re.match(r'^<(.*)>:',l) or re.match(r'^[0-9a-f]+ <(.*)>:',l)
    if m and not m.group(1).startswith('L') and '.' not in m.group(1)[:1]: cur=m.group(1); h[cur]=[hashlib.md5(),0]; continue
    if cur and l.strip():
        t=re.sub(r'//.*','',l).strip()
        h[cur][0].update(t.encode()); h[cur][1]+=1
for k,(m,n) in h.items(): print(m.hexdigest()[:12], n, k)
"
done
rm -rf $T
