import itertools, sys
sys.path.insert(0,'/root/repo/oracle')
from sim_fast2048 import _R128,_R64,_W64,_W128
def worst(groups, addr_dw, ndw, nbanks):
    w=0
    for g in groups:
        pb={}
        for l in g:
            a=addr_dw(l)
            for k in range(ndw):
                pb.setdefault((a+k)%nbanks,set()).add(a+k)
        w=max(w,max(len(v) for v in pb.values()))
    return w
def cost(groups, addr_dw, ndw, nbanks):
    # total LDS cycles = sum over groups of max bank multiplicity
    c=0
    for g in groups:
        pb={}
        for l in g:
            a=addr_dw(l)
            for k in range(ndw):
                pb.setdefault((a+k)%nbanks,set()).add(a+k)
        c+=max(len(v) for v in pb.values())
    return c
a3=lambda l:l&7
g3=lambda l:l>>3
klo=lambda l:128-l if l else 64
# exchange 1: addr(k2,a,b) complex units
def ex1_eval(addr):
    res={}
    # fwd write b64: instr k2: lane l=(a=l&7,b=l>>3)
    res['w64']=sum(cost(_W64, lambda l:2*addr(k2,l&7,l>>3),2,32) for k2 in range(16))
    res['r64']=sum(cost(_R64, lambda l:2*addr(k2,l&7,l>>3),2,64) for k2 in range(16))
    res['r128']=sum(cost(_R128, lambda l:2*addr(g3(l)+8*h,a3(l),2*c),4,64) for h in range(2) for c in range(4))
    res['w128']=sum(cost(_W128, lambda l:2*addr(g3(l)+8*h,a3(l),2*c),4,32) for h in range(2) for c in range(4))
    return res
ideal1={'w64':16*4,'r64':16*2,'r128':8*4,'w128':8*8}
best=[]
for S in range(64,132,2):
  for P in range(8,17,2):
    for xm in range(0,6):
        def addr(k2,a,b,S=S,P=P,xm=xm):
            x=[0,a&3,(a>>1)&3,k2&3,(k2>>1)&3,(a^k2)&3][xm]
            return k2*S+a*P+2*((b>>1)^x)+(b&1)
        # validity: distinct addresses
        s=set(addr(k2,a,b) for k2 in range(16) for a in range(8) for b in range(8))
        if len(s)!=1024: continue
        r=ex1_eval(addr)
        extra=sum(r[k]-ideal1[k] for k in r)
        best.append((extra,S,P,xm,r))
best.sort(key=lambda t:(t[0],t[1]))
print('ex1 ideal',ideal1)
for b in best[:8]: print(b)
# exchange 2: element (a, i) i in 0..127: i = k2+16d ; addr = a*S2 + (k2)*Q + d  with d-run swizzle
def ex2_eval(addr):
    res={}
    res['w128']=sum(cost(_W128, lambda l:2*addr(a3(l),g3(l)+8*h,2*c),4,32) for h in range(2) for c in range(4))
    res['r128']=sum(cost(_R128, lambda l:2*addr(a3(l),g3(l)+8*h,2*c),4,64) for h in range(2) for c in range(4))
    res['r64']=sum(cost(_R64, lambda l:2*addr(a,l&15,l>>4),2,64)+cost(_R64, lambda l:2*addr(a,klo(l)&15,klo(l)>>4),2,64) for a in range(8))
    res['w64']=sum(cost(_W64, lambda l:2*addr(a,l&15,l>>4),2,32)+cost(_W64, lambda l:2*addr(a,klo(l)&15,klo(l)>>4),2,32) for a in range(8))
    return res
ideal2={'w128':64,'r128':32,'r64':32,'w64':64}
best=[]
for S2 in range(128,264,2):
  for Q in range(8,17,2):
    for xm in range(0,6):
        def addr(a,k2,d,S2=S2,Q=Q,xm=xm):
            x=[0,a&3,(a>>1)&3,k2&3,(k2>>1)&3,(a^k2)&3][xm]
            return a*S2+k2*Q+2*((d>>1)^x)+(d&1)
        s=set(addr(a,k2,d) for a in range(8) for k2 in range(16) for d in range(8))
        if len(s)!=1024: continue
        r=ex2_eval(addr)
        extra=sum(r[k]-ideal2[k] for k in r)
        best.append((extra,S2,Q,xm,r))
best.sort(key=lambda t:(t[0],t[1]))
print('ex2 ideal',ideal2)
for b in best[:8]: print(b)
# current
def cur1(k2,a,b): return k2*80+a*10+b
def cur2(a,k2,d): return a*166+k2*10+d
print('current ex1',ex1_eval(cur1)); print('current ex2',ex2_eval(cur2))
