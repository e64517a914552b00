#!/usr/bin/env python3
"""How often, and when, does this compiler's v_bitop3_b32 formation get the truth table wrong?  (docs/compiler_findings.md, finding 3)

Random boolean expressions over three i32 inputs -- 2 to 5 and / or / xor / not instructions, operands drawn from everything defined
so far, so inner values are shared now and then -- go through `llc -mcpu=gfx950` in one module; the emitted instructions of each
function (v_bitop3_b32, v_and / v_or / v_xor / v_not / v_xnor / v_and_or / v_or3 / v_bfi / v_mov) are evaluated on the 8-bit truth
tables 0xf0 / 0xcc / 0xaa and compared with the expression's own table.  Classes: fused into a bitop3 or not; some inner value used
twice ("shared") or a tree.

    python3 tools/fuzz_bitop3_tables.py <seed> <functions>

Recorded (profiles/r06_code_object_rehearsal.txt): seeds 1-8 x 800 = 6 400 expressions: 12 wrong, every one of them fused AND with a
shared inner value (12 of 1 189 such); 1 356 fused trees and 3 855 unfused expressions all right.  The audit of the product
(tools/audit_bitop3.py) looks for exactly the shared shape."""
import collections
import os
import random
import re
import subprocess
import sys
import tempfile
LLC="/opt/rocm/lib/llvm/bin/llc"
rng=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
NF=int(sys.argv[2]) if len(sys.argv)>2 else 400
OPS={"and":lambda a,b:a&b,"or":lambda a,b:a|b,"xor":lambda a,b:a^b}
funcs=[]
for n in range(NF):
    k=rng.randrange(2,6)
    nodes=[("leaf","%A",0xf0),("leaf","%b",0xcc),("leaf","%X",0xaa)]
    lines=[]; uses=collections.Counter()
    for j in range(k):
        op=rng.choice(list(OPS))
        # prefer recent nodes so that the expression stays connected
        cand=list(range(len(nodes)))
        x=rng.choice(cand[-4:]); y=rng.choice(cand)
        if rng.random()<0.15:   # a NOT
            name=f"%n{j}"; lines.append(f"  {name} = xor i32 {nodes[x][1]}, -1"); nodes.append(("op",name,(~nodes[x][2])&0xff,)); uses[x]+=1
            continue
        if x==y: y=(y+1)%len(nodes)
        name=f"%n{j}"; lines.append(f"  {name} = {op} i32 {nodes[x][1]}, {nodes[y][1]}")
        nodes.append(("op",name,OPS[op](nodes[x][2],nodes[y][2])&0xff)); uses[x]+=1; uses[y]+=1
    shared=any(uses[i]>=2 and nodes[i][0]=="op" for i in range(len(nodes)))
    funcs.append((n,lines,nodes[-1][1],nodes[-1][2],shared))
ir="".join(f"define i32 @f{n}(i32 %A, i32 %b, i32 %X) {{\n"+"\n".join(l)+f"\n  ret i32 {ret}\n}}\n" for n,l,ret,_,_ in funcs)
work=tempfile.mkdtemp(prefix="b3tables_")
open(os.path.join(work,"m.ll"),"w").write(ir)
asm=subprocess.run([LLC,"-mtriple=amdgcn-amd-amdhsa","-mcpu=gfx950","-O3",os.path.join(work,"m.ll"),"-o","-"],capture_output=True,text=True).stdout
bodies=dict((int(m.group(1)),m.group(2)) for m in re.finditer(r"^f(\d+):.*?\n(.*?)s_setpc_b64",asm,re.S|re.M))
def val(tok,regs):
    if tok in regs: return regs[tok]
    if tok=="-1": return 0xff
    if tok=="0": return 0
    raise KeyError(tok)
stats=collections.Counter(); wrong=[]
for n,lines,ret,truth,shared in funcs:
    regs={"v0":0xf0,"v1":0xcc,"v2":0xaa}; ok=True; used_b3=False
    for l in bodies[n].splitlines():
        l=l.split(";")[0].strip()
        if not l or l.startswith("s_waitcnt") or l.startswith(";") or l.startswith("."): continue
        m=re.match(r"(\S+) (.*)",l); op=m.group(1); a=[x.strip() for x in m.group(2).split(",")]
        try:
            if op=="v_bitop3_b32":
                mm=re.search(r"bitop3:(\S+)",l); t=int(mm.group(1),0) if mm else 0; a[3]=a[3].split()[0]
                s0,s1,s2=(val(x,regs) for x in a[1:4]); r=0
                for bit in range(8):
                    idx=(((s0>>bit)&1)<<2)|(((s1>>bit)&1)<<1)|((s2>>bit)&1)
                    r|=((t>>idx)&1)<<bit
                regs[a[0]]=r; used_b3=True
            elif op in("v_and_b32_e32","v_or_b32_e32","v_xor_b32_e32"): regs[a[0]]=OPS[op[2:op.index("_b32")]](val(a[1],regs),val(a[2],regs))
            elif op=="v_not_b32_e32": regs[a[0]]=(~val(a[1],regs))&0xff
            elif op=="v_xnor_b32_e32": regs[a[0]]=(~(val(a[1],regs)^val(a[2],regs)))&0xff
            elif op=="v_and_or_b32": regs[a[0]]=(val(a[1],regs)&val(a[2],regs))|val(a[3],regs)
            elif op=="v_or3_b32": regs[a[0]]=val(a[1],regs)|val(a[2],regs)|val(a[3],regs)
            elif op=="v_bfi_b32": regs[a[0]]=(val(a[1],regs)&val(a[2],regs))|((~val(a[1],regs))&0xff&val(a[3],regs))
            elif op=="v_mov_b32_e32": regs[a[0]]=val(a[1],regs)
            else: ok=False; stats["skipped:"+op]+=1; break
        except KeyError: ok=False; stats["skipped:operand"]+=1; break
    if not ok: continue
    key=("bitop3" if used_b3 else "plain", "shared" if shared else "tree")
    if regs["v0"]==truth: stats[key+("right",)]+=1
    else: stats[key+("WRONG",)]+=1; wrong.append((n,lines,hex(truth),hex(regs["v0"]),bodies[n].strip().splitlines()[-2:]))
for k,v in sorted(stats.items(),key=str): print(v,k)
for w in wrong[:4]: print(w)
