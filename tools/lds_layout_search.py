import itertools
RG128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG128=RG128+[[l+32 for l in g] for g in RG128]
def conflicts_read_b128(addr_of_lane):
    worst=1
    for g in RG128:
        banks={}
        for l in g:
            a=addr_of_lane(l)
            for d in range(4):
                b=(a//4+d)%64
                banks.setdefault(b,set()).add(a//4+d)
        worst=max(worst,max(len(v) for v in banks.values()))
    return worst
def conflicts_write_b64(addr_of_lane):
    worst=1
    for g0 in range(0,64,16):
        banks={}
        for l in range(g0,g0+16):
            a=addr_of_lane(l)
            for d in range(2):
                b=(a//4+d)%32
                banks.setdefault(b,set()).add(a//4+d)
        worst=max(worst,max(len(v) for v in banks.values()))
    return worst
def conflicts_write_b128(addr_of_lane):
    worst=1
    for g0 in range(0,64,8):
        banks={}
        for l in range(g0,g0+8):
            a=addr_of_lane(l)
            for d in range(4):
                b=(a//4+d)%32
                banks.setdefault(b,set()).add(a//4+d)
        worst=max(worst,max(len(v) for v in banks.values()))
    return worst

BM=32
maps={'M1':lambda t:(t%32,t//32),'M2':lambda t:((t%8)+8*(t//64),(t%64)//8),'M3':lambda t:((t%16)+16*(t//128),(t%128)//16), 'M4':lambda t:((t%4)+4*(t//32)%32, (t%32)//4)}
rots={'none':lambda c:0,'c>>1':lambda c:(c>>1)&3,'c>>2':lambda c:(c>>2)&3,'c>>3':lambda c:(c>>3)&3,'c>>4':lambda c:(c>>4)&3,'c>>2^c>>4':lambda c:((c>>2)^(c>>4))&3, 'c>>3+c>>5':lambda c:((c>>3)+(c>>5))&3}
nch=BM//8  # chunks of 16B (8 rows bf16)
for LDT in (80,96,112,128,144):
  for rn,rot in rots.items():
    # read: lane (i,h), block base cb multiple of 32, kstep s: chunk j=2s+h
    wr=0
    for s in range(BM//16):
        for cb in (0,32,64,96):
            def ad(l,s=s,cb=cb):
                i,h=l%32,l//32; c=cb+i; j=2*s+h
                return c*LDT+((j+rot(c))%nch)*16
            wr=max(wr,conflicts_read_b128(ad))
    for mn,mp in maps.items():
        ww=0
        for wave in range(4):
            for e in range(4):
                def ad(l,wave=wave,e=e):
                    c4,rg=mp(wave*64+l); c=4*c4+e
                    return c*LDT+(((rg//2)+rot(c))%nch)*16+(rg&1)*8
                ww=max(ww,conflicts_write_b64(ad))
        if wr==1 and ww<=2:
            print("LDT",LDT,"rot",rn,"map",mn,"read",wr,"write",ww)

print("---- wide search")
def mk_map(A, nrg, ncg):
    # 256 threads: c4 = (t % A) + A * (t // (A*nrg)), rg = (t // A) % nrg ; requires A*nrg*(ncg/A) = 256
    return lambda t: ((t % A) + A * (t // (A * nrg)), (t // A) % nrg)
found=[]
for BM_,C in ((32,128),(64,64),(32,64),(64,128)):
    nrg, ncg = BM_//4, C//4
    if nrg*ncg != 256: 
        continue
    nch = BM_//8
    for LDT in range(2*BM_+16, 2*BM_+16+128, 16):
        for rn,rot in list(rots.items())+[('c>>2&1',lambda c:(c>>2)&1),('2*(c>>2&1)',lambda c:2*((c>>2)&1)),('c>>3&1',lambda c:(c>>3)&1)]:
            wr=0
            for s_ in range(BM_//16):
                for cb in range(0,C,32):
                    def ad(l,s_=s_,cb=cb):
                        i,h=l%32,l//32; c=cb+i; j=2*s_+h
                        return c*LDT+((j+rot(c))%nch)*16
                    wr=max(wr,conflicts_read_b128(ad))
            if wr>1: continue
            for A in (1,2,4,8,16,32):
                if A>ncg: continue
                mp=mk_map(A,nrg,ncg)
                ww=0
                for wave in range(4):
                    for e in range(4):
                        def ad(l,wave=wave,e=e):
                            c4,rg=mp(wave*64+l); c=4*c4+e
                            return c*LDT+(((rg//2)+rot(c))%nch)*16+(rg&1)*8
                        ww=max(ww,conflicts_write_b64(ad))
                if ww==1:
                    print("BM",BM_,"C",C,"LDT",LDT,"rot",rn,"A",A,"read",wr,"write",ww)

print("---- verify chosen S2 layouts")
def pi(m): 
    return ((m>>2)^(m&3)) + 8*((m>>1)&1) + 16*(m&1)
assert sorted(pi(m) for m in range(32))==list(range(32))
for Co,BM_ in ((128,32),(64,32),(128,64),(64,64)):
    LDR=2*Co+16
    nrg,ncg=BM_//4,Co//4
    nunits=nrg*ncg
    # writes: for each unit-slot u (0..nunits/256), wave, row j: 8B at slot(row)*LDR + 8*c4
    ww=0
    for u in range(max(1,nunits//256)):
        for wave in range(4):
            for j in range(4):
                def ad(l,wave=wave,j=j,u=u):
                    p=wave*64+l+256*u
                    if p>=nunits: return 10**9+l*64   # idle lanes: distinct
                    c4=(p%2)+2*(p//(2*nrg)); rg=(p//2)%nrg
                    m=4*rg+j
                    slot=32*(m//32)+pi(m%32)
                    return slot*LDR+8*c4
                ww=max(ww,conflicts_write_b64(ad))
    wr=0
    for q in range(Co//16):
        for wr_ in range(BM_//32):
            def ad(l,q=q,wr_=wr_):
                i,h=l%32,l//32
                return (32*wr_+pi(i))*LDR+(16*q+8*h)*2
            wr=max(wr,conflicts_read_b128(ad))
    print("DYR Co",Co,"BM",BM_,"write",ww,"read",wr)
# XR: [C][BM+4] floats; write b128 per channel e: c*LDX + 16*rg ; read: lane i: c=cb+i: c*LDX + (8g+4h)*4 + wr*32*4
for Ci,BM_ in ((128,32),(64,64)):
    LDX=4*(BM_+4); nrg,ncg=BM_//4,Ci//4
    ww=0
    for wave in range(4):
        for e in range(4):
            def ad(l,wave=wave,e=e):
                p=wave*64+l; c4=(p%2)+2*(p//(2*nrg)); rg=(p//2)%nrg
                return (4*c4+e)*LDX+16*rg
            ww=max(ww,conflicts_write_b128(ad))
    wr=0
    for g_ in range(4):
        for cb in range(0,Ci,32):
            def ad(l,g_=g_,cb=cb):
                i,h=l%32,l//32
                return (cb+i)*LDX+(8*g_+4*h)*4
            wr=max(wr,conflicts_read_b128(ad))
    print("XR Ci",Ci,"BM",BM_,"write",ww,"read",wr)
# DYT/XT with A=2, LDT=2*BM+16 no rotation, all (C,BM)
for C,BM_ in ((128,32),(64,64),(128,64),(64,32)):
    LDT=2*BM_+16; nrg,ncg=BM_//4,C//4; nunits=nrg*ncg
    ww=0
    for u in range(max(1,nunits//256)):
        for wave in range(4):
            for e in range(4):
                def ad(l,wave=wave,e=e,u=u):
                    p=wave*64+l+256*u
                    if p>=nunits: return 10**9+l*64
                    c4=(p%2)+2*(p//(2*nrg)); rg=(p//2)%nrg
                    return (4*c4+e)*LDT+8*rg
                ww=max(ww,conflicts_write_b64(ad))
    wr=0
    for s_ in range(BM_//16):
        for cb in range(0,C,32):
            def ad(l,s_=s_,cb=cb):
                i,h=l%32,l//32
                return (cb+i)*LDT+(2*s_+h)*16
            wr=max(wr,conflicts_read_b128(ad))
    print("DYT C",C,"BM",BM_,"LDT",LDT,"write",ww,"read",wr)

print("---- A=8 mapping (global loads: a quarter-wave covers 2 rows x 128 B = whole lines)")
def mapA(A, nrg):
    return lambda p: ((p % A) + A * (p // (A * nrg)), (p // A) % nrg)
for C,BM_ in ((128,32),(64,64),(128,64),(64,32)):
    nrg,ncg=BM_//4,C//4; nunits=nrg*ncg; nch=BM_//8
    mp=mapA(8,nrg)
    for LDT in range(2*BM_+16, 2*BM_+16+160, 16):
        for rn,rot in list(rots.items()):
            wr=0
            for s_ in range(BM_//16):
                for cb in range(0,C,32):
                    def ad(l,s_=s_,cb=cb):
                        i,h=l%32,l//32; c=cb+i; j=2*s_+h
                        return c*LDT+((j+rot(c))%nch)*16
                    wr=max(wr,conflicts_read_b128(ad))
            if wr>1: continue
            ww=0
            for u in range(max(1,nunits//256)):
                for wave in range(4):
                    for e in range(4):
                        def ad(l,wave=wave,e=e,u=u):
                            p=wave*64+l+256*u
                            if p>=nunits: return None
                            c4,rg=mp(p); c=4*c4+e
                            return c*LDT+(((rg//2)+rot(c))%nch)*16+(rg&1)*8
                        # idle lanes excluded
                        worst=1
                        for g0 in range(0,64,16):
                            banks={}
                            for l in range(g0,g0+16):
                                a_=ad(l)
                                if a_ is None: continue
                                for d in range(2):
                                    banks.setdefault((a_//4+d)%32,set()).add(a_//4+d)
                            if banks: worst=max(worst,max(len(v) for v in banks.values()))
                        ww=max(ww,worst)
            if ww==1: print("DYT C",C,"BM",BM_,"LDT",LDT,"rot",rn,"read",wr,"write",ww)
# DYR natural order with A=8
for Co,BM_ in ((128,32),(64,32),(128,64),(64,64)):
    LDR=2*Co+16; nrg,ncg=BM_//4,Co//4; nunits=nrg*ncg; mp=mapA(8,nrg)
    ww=0
    for u in range(max(1,nunits//256)):
        for wave in range(4):
            for j in range(4):
                worst=1
                for g0 in range(0,64,16):
                    banks={}
                    for l in range(g0,g0+16):
                        p=wave*64+l+256*u
                        if p>=nunits: continue
                        c4,rg=mp(p); a_=(4*rg+j)*LDR+8*c4
                        for d in range(2): banks.setdefault((a_//4+d)%32,set()).add(a_//4+d)
                    if banks: worst=max(worst,max(len(v) for v in banks.values()))
                ww=max(ww,worst)
    wr=0
    for q in range(Co//16):
        for wr_ in range(BM_//32):
            def ad(l,q=q,wr_=wr_):
                i,h=l%32,l//32
                return (32*wr_+i)*LDR+(16*q+8*h)*2
            wr=max(wr,conflicts_read_b128(ad))
    print("DYR natural Co",Co,"BM",BM_,"write",ww,"read",wr)
# XR with A=8: [C][LDX] fp32, chunk rotation f(c4)
for Ci,BM_ in ((128,32),(64,64)):
    nrg,ncg=BM_//4,Ci//4; mp=mapA(8,nrg); nchx=BM_//4
    for LDXf in range(BM_+4, BM_+4+36, 4):
        LDX=4*LDXf
        for fn,f in (('none',lambda c4:0),('c4',lambda c4:c4),('c4>>1',lambda c4:c4>>1),('2*c4',lambda c4:2*c4)):
            ww=0
            for wave in range(4):
                for e in range(4):
                    def ad(l,wave=wave,e=e):
                        p=wave*64+l; c4,rg=mp(p)
                        return (4*c4+e)*LDX+16*((rg+f(c4))%nchx)
                    ww=max(ww,conflicts_write_b128(ad))
            wr=0
            for g_ in range(4):
                for wr_ in range(BM_//32):
                    for cb in range(0,Ci,32):
                        def ad(l,g_=g_,cb=cb,wr_=wr_):
                            i,h=l%32,l//32; c=cb+i; chunk=(wr_*32+8*g_+4*h)//4
                            return c*LDX+16*((chunk+f(c//4))%nchx)
                        wr=max(wr,conflicts_read_b128(ad))
            if ww==1 and wr==1: print("XR Ci",Ci,"BM",BM_,"LDX floats",LDXf,"rot",fn,"write",ww,"read",wr)

print("---- A=8: best (read, write) conflict levels for the transposed pieces")
for C,BM_ in ((128,32),(64,64),(128,64),(64,32)):
    nrg,ncg=BM_//4,C//4; nunits=nrg*ncg; nch=BM_//8
    mp=mapA(8,nrg)
    best=[]
    for LDT in range(2*BM_+16, 2*BM_+16+96, 16):
        for rn,rot in list(rots.items()):
            wr=0
            for s_ in range(BM_//16):
                for cb in range(0,C,32):
                    def ad(l,s_=s_,cb=cb):
                        i,h=l%32,l//32; c=cb+i; j=2*s_+h
                        return c*LDT+((j+rot(c))%nch)*16
                    wr=max(wr,conflicts_read_b128(ad))
            ww=0
            for u in range(max(1,nunits//256)):
                for wave in range(4):
                    for e in range(4):
                        worst=1
                        for g0 in range(0,64,16):
                            banks={}
                            for l in range(g0,g0+16):
                                p=wave*64+l+256*u
                                if p>=nunits: continue
                                c4,rg=mp(p); c=4*c4+e
                                a_=c*LDT+(((rg//2)+rot(c))%nch)*16+(rg&1)*8
                                for d in range(2): banks.setdefault((a_//4+d)%32,set()).add(a_//4+d)
                            if banks: worst=max(worst,max(len(v) for v in banks.values()))
                        ww=max(ww,worst)
            best.append((wr+ww,wr,ww,LDT,rn))
    best.sort()
    print("C",C,"BM",BM_,best[:4])

print("---- A=8, BM=64: conflict levels at LDT=144 and XR LDX=68 (LDS budget of the Co=128, Ci=64 shape)")
for C,BM_ in ((64,64),(128,64)):
    nrg,ncg=BM_//4,C//4; nunits=nrg*ncg; nch=BM_//8; mp=mapA(8,nrg); LDT=144
    for rn,rot in list(rots.items()):
        wr=0
        for s_ in range(BM_//16):
            for cb in range(0,C,32):
                def ad(l,s_=s_,cb=cb):
                    i,h=l%32,l//32; c=cb+i; j=2*s_+h
                    return c*LDT+((j+rot(c))%nch)*16
                wr=max(wr,conflicts_read_b128(ad))
        ww=0
        for u in range(max(1,nunits//256)):
            for wave in range(4):
                for e in range(4):
                    worst=1
                    for g0 in range(0,64,16):
                        banks={}
                        for l in range(g0,g0+16):
                            p=wave*64+l+256*u
                            c4,rg=mp(p); c=4*c4+e
                            a_=c*LDT+(((rg//2)+rot(c))%nch)*16+(rg&1)*8
                            for d in range(2): banks.setdefault((a_//4+d)%32,set()).add(a_//4+d)
                        worst=max(worst,max(len(v) for v in banks.values()))
                    ww=max(ww,worst)
        print("C",C,"LDT 144 rot",rn,"read",wr,"write",ww)
Ci,BM_=64,64
nrg=BM_//4; mp=mapA(8,nrg); nchx=BM_//4
for LDXf in (68,72,76):
    LDX=4*LDXf
    for fn,f in (('none',lambda c4:0),('c4',lambda c4:c4),('c4>>1',lambda c4:c4>>1),('2*c4',lambda c4:2*c4)):
        ww=0
        for wave in range(4):
            for e in range(4):
                def ad(l,wave=wave,e=e):
                    p=wave*64+l; c4,rg=mp(p)
                    return (4*c4+e)*LDX+16*((rg+f(c4))%nchx)
                ww=max(ww,conflicts_write_b128(ad))
        wr=0
        for g_ in range(4):
            for wr_ in range(2):
                for cb in range(0,Ci,32):
                    def ad(l,g_=g_,cb=cb,wr_=wr_):
                        i,h=l%32,l//32; c=cb+i; chunk=(wr_*32+8*g_+4*h)//4
                        return c*LDX+16*((chunk+f(c//4))%nchx)
                    wr=max(wr,conflicts_read_b128(ad))
        print("XR BM=64 LDX",LDXf,"rot",fn,"write",ww,"read",wr)
