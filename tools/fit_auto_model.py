"""Replays AUTO's cost model (csrc/spmm.hip spmm_auto_cost + csrc/spmm_rowsplit.hip rowsplit_est_us / rowsplit_panels /
rowsplit_segments, restated here) over the MEASURED map (profiles/r04_auto_map.json, tools/auto_map.py) with other
constants: for every shape above 50 us, which kernel the model would pick and how far the pick's measured time is from the
best measured candidate.  How round 4's constants (6 us per panel launch, the row-group rates, 0.005 ns per (row, slab,
panel) of the planned sweep) were found without a GPU run per try.   python tools/fit_auto_model.py [map.json]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_args = [a for a in sys.argv[1:] if not a.startswith("-")] if __name__ == "__main__" else []
d=json.load(open(_args[0] if _args else os.path.join(ROOT, "profiles", "r04_auto_map.json")))
pts=d['spmm']
def G_of(n,sz):
    vec=16//sz
    return 8 if n<=8*vec else 16 if n<=16*vec else 32 if n<=32*vec else 64
def panels(m,n,K,sz,avg,P):
    b=K*n*sz
    if b<7e6: return 1
    if m*((n*sz+1023)//1024) < P['round']: return 1
    p=int((b+3e6-1)//3e6)
    rb=n*sz
    per=96.0 if rb<=128 else 64.0 if rb<=256 else 32.0
    p=min(p,int(avg/per),32)
    return max(p,1)
def segments(m,n,sz,avg,P):
    vec=16//sz; G=G_of(n,sz)
    lim={8:P['l8'],16:P['l16'],32:P['l32']}.get(G,0)
    if G<64 and n%vec==0 and avg<=lim and m*G>=8192*64: return 0
    W=64*vec; passes=(n+W-1)//W; S=1
    while S<8 and m*S*passes<3000 and avg/(2*S)>=64: S*=2
    return S
def cost(m,n,K,nnz,sz,keep,P):
    avg=nnz/m; W=64*(16//sz); passes=(n+W-1)//W
    cpl=128//sz; slabs=(n+cpl-1)//cpl; b=K*n*sz
    def rate(pb,l2,mall):
        hit=min(1.0,4*1048576.0/pb)
        return 1e6/(hit/l2+(1-hit)/mall)
    pn=panels(m,n,K,sz,avg,P)
    one_line=n*sz<=128
    lanes=17.0 if one_line else 28.0
    S=segments(m,n,sz,avg/pn,P)
    G=G_of(n,sz)
    mall=8.5
    rowc=0.2e-3*m*pn*passes
    if S==0:
        lanes=P['rg8'] if G==8 else P['rg16']
        rowc={8:P['c8'],16:P['c16'],32:P['c32']}[G]*m*pn
        mall=P['rgmall']
    cb=m*n*sz
    ctr=(2.0*pn-2.0)*cb/5e6 if cb>128e6 else 0.0
    rs=nnz*n*sz/rate(b/pn,lanes,mall)+rowc+4.0*pn+4.0+ctr
    pp=max(1.0,math.ceil(K*128.0/2.5e6))
    pairs=math.ceil(m/64.0)*slabs; fill=pairs/(pairs+P['fillk'])
    sw=(19.0 if pp>1 else 23.5)*fill
    pl=nnz*slabs*128.0/rate(K*128.0/pp,sw,8.5*fill)+P['prow']*m*slabs*pp+15.0+(0.0 if keep else 36.0+6e-6*nnz)
    return rs,pl,pn,S
def evaluate(P,verbose=False):
    worst=[0,0]; over=[0,0]; rows=[]
    for p in pts:
        ms=p['ms']; m,n,K,npr=p['m'],p['n'],p['K'],p['per_row']; sz=8 if p['dtype']=='f64' else 4; nnz=m*npr
        if min(v for v in ms.values() if v)<0.05: continue
        for ki,keep in enumerate((False,True)):
            # preconditions of spmm_auto_algo
            if nnz*n < (1<<22): continue
            rs,pl,pn,S=cost(m,n,K,nnz,sz,keep,P)
            if m<32768: choice='rs'
            else: choice='pl' if pl<rs else 'rs'
            if choice=='pl': t=ms['planned_kept'] if keep else ms['planned_rebuilt']
            else:
                if pn==1: t=ms.get('rowsplit_row_groups') if S==0 else (ms['rowsplit_wave_per_row'] if S==1 else ms['rowsplit_one_panel'])
                else: t=ms['rowsplit']
                if t is None: t=ms['rowsplit_one_panel']
            cands=['rowwave','slab','rowsplit','rowsplit_one_panel','rowsplit_wave_per_row','rowsplit_row_groups','planned_kept' if keep else 'planned_rebuilt']
            best=min(ms[c] for c in cands if ms.get(c))
            r=t/best
            worst[ki]=max(worst[ki],r); over[ki]+=r>1.25
            if verbose and r>1.15: rows.append((round(r,3),m,K,npr,n,p['layout'],p['dtype'],'keep' if keep else 'one',choice,pn,S,round(rs),round(pl),round(t*1e3),round(best*1e3)))
    if verbose:
        for r in sorted(rows,reverse=True): print(r)
    return worst,over
base=dict(round=4096,l8=56,l16=40,l32=24,rg8=17.0,rg16=28.0,c8=0.2e-3/8,c16=0.2e-3/4,c32=0.2e-3/2,rgmall=8.5,fillk=2400.0,prow=0.01e-3)

def cost2(m,n,K,nnz,sz,keep,P):
    avg=nnz/m; W=64*(16//sz); passes=(n+W-1)//W
    cpl=128//sz; slabs=(n+cpl-1)//cpl; b=K*n*sz
    def rate(pb,l2,mall):
        hit=min(1.0,P['H']/pb)
        return 1e6/(hit/l2+(1-hit)/mall)
    G=G_of(n,sz); one_line=n*sz<=128
    cb=m*n*sz
    def rs_est(pn):
        lanes=17.0 if one_line else 28.0*(min(1.0, n*sz/(G*16.0)) if G<64 else 1.0)
        S=segments(m,n,sz,avg/pn,P); mall=8.5
        rowc=0.2e-3*m*pn*passes
        if S==0:
            lanes=P['rg8'] if G==8 else P['rg16']
            rowc={8:P['c8'],16:P['c16'],32:P['c32']}[G]*m*pn
            mall=P['rgmall']
        ctr=(2.0*pn-2.0)*cb/P['cbw'] if cb>128e6 else 0.0
        return nnz*n*sz/rate(b/pn,lanes,mall)+rowc+P['PP']*pn+4.0+ctr, S
    ph=panels(m,n,K,sz,avg,P)
    e1,S1=rs_est(1); eh,Sh=rs_est(ph)
    if ph>1 and eh<e1: rs,pn,S=eh,ph,Sh
    else: rs,pn,S=e1,1,S1
    pp=max(1.0,math.ceil(K*128.0/2.5e6))
    pairs=math.ceil(m/64.0)*slabs; fill=pairs/(pairs+P['fillk'])
    sw=(19.0 if pp>1 else 23.5)*fill
    def prate(pb,l2,mall):
        hit=min(1.0,4*1048576.0/pb)
        return 1e6/(hit/l2+(1-hit)/mall)
    pl=nnz*slabs*128.0/prate(K*128.0/pp,sw,8.5*fill)+P['prow']*m*slabs*pp+15.0+(0.0 if keep else 36.0+6e-6*nnz)
    return rs,pl,pn,S
cost_orig=cost
def run(P,verbose=False):
    global cost
    cost=cost2
    r=evaluate(P,verbose)
    cost=cost_orig
    return r
SHIPPED = dict(base, full=3.5e6, H=4 * 1048576.0, PP=6.0, cbw=5e6, prow=0.005e-3, rg8=18.0, rg16=21.0, c8=0.0, c16=0.07e-3,
               c32=0.15e-3, rgmall=6.5)
if __name__ == '__main__':
    print("shipped constants: worst ratio [one-shot, plan kept], points above 1.25:", run(SHIPPED, True))
    if "--grid" in sys.argv:
        for H in (4, 5, 6):
            for PP in (4, 6, 8):
                for prow in (0.01e-3, 0.005e-3):
                    P = dict(SHIPPED, H=H * 1048576.0, PP=PP, prow=prow)
                    w, o = run(P)
                    print(H, PP, prow, [round(x, 3) for x in w], o)
