"""Lists the s_waitcnt instructions of one kernel that retire loads whose destination registers are next WRITTEN rather than read -- the signature of a
load with a dead destination half: the register allocator reuses the dead registers, and the overwrite must wait for the load to land (a full memory round trip
in front of whatever comes next).  Found the serialised terrain gather of round 5 (profiles/r05s_gather_dead_pair_ab.txt).
usage: llvm-objdump -d --no-show-raw-insn <code object> > k.s ; python tools/isa_waw_waits.py k.s <kernel name prefix> vm|lgkm
(a linear scan: it does not follow branches, so hits inside loop epilogues need a look at the code; accumulating ops such as v_fmac read their destination
and show up as false positives)"""
import re, sys
def kern(path, name):
    out=[]; on=False
    for line in open(path):
        m=re.match(r'^[0-9a-f]{16} <(.+)>:',line)
        if m: on = m.group(1).startswith(name); continue
        if on and line.strip(): out.append(line.strip().split("//")[0].strip())
    return out
def regs(tok):
    tok=tok.strip().rstrip(',')
    m=re.match(r'^-?\|?v\[(\d+):(\d+)\]\|?$',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'^-?\|?v(\d+)\|?$',tok)
    if m: return {int(m.group(1))}
    return set()
def parse(l):
    parts=l.split(None,1)
    op=parts[0]; ops=[t for t in (parts[1].split(',') if len(parts)>1 else [])]
    ops=[o.strip() for o in ops]
    return op,ops
path,name,kind=sys.argv[1],sys.argv[2],sys.argv[3]   # kind: vm or lgkm
k=kern(path,name)
out=[]  # outstanding (idx, destregs)
for i,l in enumerate(k):
    op,ops=parse(l)
    isload = (op.startswith(("global_load","buffer_load","scratch_load")) if kind=="vm" else op.startswith("ds_read"))
    if isload:
        out.append((i,regs(ops[0])))
        continue
    if op=="s_barrier" or op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc"):
        pass
    m=re.search(r'vmcnt\((\d+)\)',l) if kind=="vm" else re.search(r'lgkmcnt\((\d+)\)',l)
    if op=="s_waitcnt" and m:
        n=int(m.group(1))
        retired=out[:max(0,len(out)-n)] if n<len(out) else []
        out=out[len(out)-n:] if n<len(out) else out
        if not retired: continue
        rr=set().union(*[r for _,r in retired])
        # look ahead
        reads=False; writes=None
        for j in range(i+1,min(len(k),i+6)):
            o2,p2=parse(k[j])
            if o2=="s_waitcnt": break
            srcs=set().union(*[regs(t) for t in p2[1:]]) if len(p2)>1 else set()
            # stores / ds_write have all sources
            if o2.startswith(("global_store","ds_write","buffer_store","scratch_store")): srcs=set().union(*[regs(t) for t in p2])
            if srcs & rr: reads=True; break
            d=regs(p2[0]) if p2 else set()
            if d & rr and not o2.startswith(("global_store","ds_write")): writes=(j,k[j]); break
        if writes and not reads:
            print(f"WAW-suspect wait at {i}: '{l}' retiring loads {[x for x,_ in retired][:4]}.. then {writes[0]}: {writes[1][:70]}")
