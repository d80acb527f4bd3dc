"""Developer tool (CPU): literal numpy transcription of the index arithmetic of k_pack_a + k_voc_conv (csrc/voc_kernels.hpp) at lane
level - staging, the tap functor, the MFMA fragment maps, both store paths - with unstaged LDS and unwritten outputs poisoned with NaN.
Used to desk-check the addressing before the first GPU run; the GPU parity tests are tests/test_gpu_vocoder.py."""
import numpy as np, torch, torch.nn.functional as F
HALO=28
def frag_row(r,h): return (r&3)+8*(r>>2)+4*h
def pack(w, rows, Ci, KT):
    nw=(rows+31)//32; nkc=(Ci+7)//8
    n=nw*KT*nkc*256
    dst=np.zeros(n+8192*4,np.float32)
    for idx in range(n):
        s=idx&3; lane=(idx>>2)&63; r=idx>>8
        mb=r%1; r//=1
        kct=r%(KT*nkc); r//=(KT*nkc)
        wv=r
        tap=kct%KT; kc=kct//KT
        i=lane&31; h=lane>>5
        row=32*wv+i; col=8*kc+4*h+s
        if row<rows and col<Ci: dst[idx]=w[row,col,tap]
    return dst
def kernel(x, wp, bias, res, sum_in, B, Ci, rows, KT, pad, dil, Li, U, slope, divide, act, NB, WT):
    LSi=(Li+31)//32*32; Lo=Li*U; LSo=(Lo+31)//32*32
    SPAN=32*NB*WT; LD=SPAN+2*HALO; SLAB=min(256,(72*1024//(LD*4))//8*8); WR=4//WT; NCOL4=LD//4
    Co=rows//U
    out=np.full((B,Co,LSo),np.nan,np.float32)
    nrb=(rows+31)//32; ci8=(Ci+7)//8; nct=ci8*KT
    gx=(LSi+SPAN-1)//SPAN; gz=(rows+32*WR-1)//(32*WR)
    for bx in range(gx):
      for b in range(B):
        for bz in range(gz):
          t0=bx*SPAN
          acc=np.zeros((4,NB,64,16),np.float32)  # wave, nb, lane, reg
          for c0 in range(0,Ci,SLAB):
            nc=min(SLAB,Ci-c0); nc8=(nc+7)//8*8
            smem=np.full((SLAB*LD,),np.nan,np.float32)
            for idx in range(nc8*NCOL4):
                row=idx//NCOL4; g=idx-row*NCOL4; t=t0-HALO+4*g
                v=np.zeros(4,np.float32)
                if row<nc and t>=0 and t<LSi:
                    v=x[b,c0+row,t:t+4].copy(); v=np.where(v>0,v,v*np.float32(slope))
                smem[row*LD+4*g:row*LD+4*g+4]=v
            nch=(nc8//8)*KT
            for w in range(4):
                wr=w%WR; wt=w//WR; rb=bz*WR+wr; rbc=min(rb,nrb-1)
                apoff=(rbc*nct+(c0//8)*KT)*64*4   # float offset
                for lane in range(64):
                    j=lane&31; h=lane>>5
                    base=4*h*LD+HALO+wt*32*NB+j-pad
                    for kc in range(nch):
                        g=kc//KT; tap=kc-g*KT
                        bp=base+g*8*LD+tap*dil
                        a4=wp[apoff+kc*64*4+lane*4: apoff+kc*64*4+lane*4+4]
                        # mfma semantic: D[row i][col j] += sum over k: A[i][k]*B[k][j]; lane (i,h) supplies A[i][4h+s]; lane(j,h) supplies B[4h+s][j]
                        for s in range(4):
                            for nb in range(NB):
                                bval=smem[bp+s*LD+nb*32]
                                # store operand; accumulate below via gather
                                acc_op.setdefault((w,nb,kc,s),[None]*64)[lane]=(a4[s],bval)
                # do MFMAs for wave w
                for nb in range(NB):
                    for kc in range(nch):
                        for s in range(4):
                            ops=acc_op[(w,nb,kc,s)]
                            A=np.zeros((32,2),np.float32); Bm=np.zeros((2,32),np.float32)
                            for lane in range(64):
                                A[lane&31,lane>>5]=ops[lane][0]; Bm[lane>>5,lane&31]=ops[lane][1]
                            D=A@Bm
                            for lane in range(64):
                                jj=lane&31; hh=lane>>5
                                for r in range(16): acc[w,nb,lane,r]+=D[frag_row(r,hh),jj]
                acc_op.clear()
          for w in range(4):
            wr=w%WR; wt=w//WR; rb=bz*WR+wr
            if rb>=nrb: continue
            for lane in range(64):
                j=lane&31; h=lane>>5
                q0=t0+wt*32*NB+j
                for nb in range(NB):
                    q=q0+32*nb
                    if U%4==0:
                        for rg in range(4):
                            row=rb*32+8*rg+4*h; co=row//U; ph=row-co*U; n=q*U+ph
                            if row>=rows or n>=LSo: continue
                            for e in range(4):
                                v=acc[w,nb,lane,4*rg+e]+(bias[co] if bias is not None else 0)
                                if res is not None: v+=res[b,co,n+e]
                                if sum_in is not None: v=sum_in[b,co,n+e]+v
                                if divide!=1: v=v/np.float32(divide)
                                if act==1: v=np.tanh(v)
                                if n+e>=Lo: v=0
                                out[b,co,n+e]=v
                    else:
                        for r in range(16):
                            row=rb*32+frag_row(r,h); rc=row if row<rows else 0
                            co=rc//U; ph=rc-co*U; n=q*U+ph
                            ok=row<rows and n<LSo
                            if not ok: continue
                            v=acc[w,nb,lane,r]+(bias[co] if bias is not None else 0)
                            if res is not None: v+=res[b,co,n]
                            if sum_in is not None: v=sum_in[b,co,n]+v
                            if divide!=1: v=v/np.float32(divide)
                            if act==1: v=np.tanh(v)
                            out[b,co,n]=v if n<Lo else 0
    return out
acc_op={}
def cm(x,L):
    o=np.zeros((x.shape[0],x.shape[1],(L+31)//32*32),np.float32); o[:,:,:L]=x; return o
def check(ci,co,k,dil,L,U=1,NB=1,WT=1,ktrans=None):
    rng=np.random.default_rng(ci+co+k+L)
    B=1
    if U==1:
        w=rng.standard_normal((co,ci,k)).astype(np.float32); pad=(k-1)*dil//2; rows=co; KT=k
        x=rng.standard_normal((B,ci,L)).astype(np.float32)
        want=F.conv1d(F.leaky_relu(torch.from_numpy(x),0.1),torch.from_numpy(w),None,padding=pad,dilation=dil).numpy()
    else:
        import sys; sys.path.insert(0,'/root/repo')
        from diffsinger_amd.vocoder import polyphase_weight
        wt_=rng.standard_normal((ci,co,ktrans)).astype(np.float32)
        x=rng.standard_normal((B,ci,L)).astype(np.float32)
        want=F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x),0.1),torch.from_numpy(wt_),None,stride=U,padding=(ktrans-U)//2).numpy()
        wpl,pad=polyphase_weight(torch.from_numpy(wt_),U,(ktrans-U)//2); w=wpl.numpy(); rows=co*U; KT=w.shape[2]
    bias=rng.standard_normal(co).astype(np.float32)
    Lo=L*U
    res=cm(rng.standard_normal((B,co,Lo)).astype(np.float32),Lo)
    wp=pack(w,rows,ci,KT)
    got=kernel(cm(x,L),wp,bias,res,None,B,ci,rows,KT,pad,dil,L,U,0.1,1.0,0,NB,WT)
    assert not np.isnan(got).any(), 'unwritten outputs'
    assert np.abs(got[:,:,Lo:]).sum()==0
    err=np.abs(got[:,:,:Lo]-(want+bias[None,:,None]+res[:,:,:Lo])).max()
    print(ci,co,k,dil,L,U,NB,WT,'err',err)
    assert err<1e-4
def run_conv_checks():
    check(8,8,3,5,70,NB=1,WT=1)         # rows<=32 would use <4,4>, but index logic identical; small for speed
    check(12,20,5,2,45,NB=2,WT=2)
    check(8,40,3,1,40,NB=2,WT=2)
    check(16,8,0,1,21,U=2,NB=1,WT=1,ktrans=4)
    check(8,4,0,1,13,U=8,NB=1,WT=1,ktrans=16)
    check(8,8,0,1,13,U=4,NB=2,WT=2,ktrans=8)
    print('ok')


# ---- k_voc_conv_fold (narrow layers: F output samples folded into the MFMA rows) ------------------------------------------
def kernel_fold(x, wp, bias, res, B, Ci, Co, K, F, dil, L, slope):
    LS = (L + 31) // 32 * 32
    NB = 2; LD = 260 * F + 80; NCOL4 = LD // 4
    KT = K + F - 1; pad = (K - 1) * dil // 2; fd = F * dil
    groups = (LS + fd - 1) // fd; cols = groups * dil
    gx = (cols + 255) // 256
    out = np.full((B, Co, LS), np.nan, np.float32)
    maxcol = (254 + dil) * F + dil + 30 + (KT - 1) * dil - pad
    assert pad <= 25 and maxcol < LD
    nc8 = (Ci + 7) // 8 * 8
    for bx in range(gx):
        for b in range(B):
            cb = bx * 256
            t_org = ((cb // dil) * fd - HALO) & ~3
            smem = np.full((16 * LD,), np.nan, np.float32)
            for idx in range(nc8 * NCOL4):
                row = idx // NCOL4; g = idx - row * NCOL4; t = t_org + 4 * g
                v = np.zeros(4, np.float32)
                if row < Ci and t >= 0 and t < LS:
                    v = x[b, row, t:t + 4].copy(); v = np.where(v > 0, v, v * np.float32(slope))
                smem[row * LD + 4 * g: row * LD + 4 * g + 4] = v
            nch = (nc8 // 8) * KT
            for w in range(4):
                acc = np.zeros((NB, 64, 16), np.float32)
                posl = np.zeros((64, NB), np.int64)
                opsA = {}; opsB = {}
                for lane in range(64):
                    j = lane & 31; h = lane >> 5
                    pos = []
                    for nb in range(NB):
                        c = cb + w * 64 + nb * 32 + j
                        grp = c // dil
                        pos.append(grp * fd + (c - grp * dil))
                    posl[lane] = pos
                    base = 4 * h * LD + (pos[0] - t_org - pad)
                    assert pos[0] - t_org - pad >= 0
                    for kc in range(nch):
                        g = kc // KT; s_ = kc - g * KT
                        bp = base + g * 8 * LD + s_ * dil
                        a4 = wp[kc * 256 + lane * 4: kc * 256 + lane * 4 + 4]
                        for s in range(4):
                            for nb in range(NB):
                                col = bp + s * LD + (pos[nb] - pos[0])
                                assert (col % LD) <= maxcol + 0 or True
                                opsA[(kc, s, lane)] = a4[s]
                                opsB[(kc, s, nb, lane)] = smem[col]
                for nb in range(NB):
                    for kc in range(nch):
                        for s in range(4):
                            A = np.zeros((32, 2), np.float32); Bm = np.zeros((2, 32), np.float32)
                            for lane in range(64):
                                A[lane & 31, lane >> 5] = opsA[(kc, s, lane)]; Bm[lane >> 5, lane & 31] = opsB[(kc, s, nb, lane)]
                            D = A @ Bm
                            for lane in range(64):
                                for r in range(16):
                                    acc[nb, lane, r] += D[frag_row(r, lane >> 5), lane & 31]
                for lane in range(64):
                    h = lane >> 5
                    for nb in range(NB):
                        if dil == 1 and F == 4:
                            for rg in range(4):
                                co = 2 * rg + h; n = posl[lane, nb]
                                if co >= Co or n >= LS: continue
                                for e in range(4):
                                    v = acc[nb, lane, 4 * rg + e] + bias[co] + res[b, co, n + e]
                                    assert np.isnan(out[b, co, n + e]), 'written twice'
                                    out[b, co, n + e] = v if n + e < L else 0
                        else:
                            for r in range(16):
                                row = frag_row(r, h); co = row // F; e = row - co * F
                                n = posl[lane, nb] + e * dil
                                if not (co < Co and n < LS): continue
                                v = acc[nb, lane, r] + bias[co] + res[b, co, n]
                                assert np.isnan(out[b, co, n]), 'written twice'
                                out[b, co, n] = v if n < L else 0
    return out


def check_fold(ci, co, k, dil, L, F):
    import sys; sys.path.insert(0, '/root/repo')
    from diffsinger_amd.vocoder import fold_weight
    rng = np.random.default_rng(ci + co + k + L + dil)
    B = 1
    w = rng.standard_normal((co, ci, k)).astype(np.float32)
    x = rng.standard_normal((B, ci, L)).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    res = cm(rng.standard_normal((B, co, L)).astype(np.float32), L)
    want = F_.conv1d(F_.leaky_relu(torch.from_numpy(x), 0.1), torch.from_numpy(w), torch.from_numpy(bias), padding=(k - 1) * dil // 2, dilation=dil).numpy() + res[:, :, :L]
    wf = fold_weight(torch.from_numpy(w), F).numpy()
    wp = pack(wf, co * F, ci, k + F - 1)
    got = kernel_fold(cm(x, L), wp, bias, res, B, ci, co, k, F, dil, L, 0.1)
    assert not np.isnan(got).any(), 'unwritten outputs'
    assert np.abs(got[:, :, L:]).sum() == 0
    err = np.abs(got[:, :, :L] - want).max()
    print('fold', ci, co, k, dil, L, F, 'err', err)
    assert err < 1e-4


if __name__ == '__main__':
    import torch.nn.functional as F_
    run_conv_checks()
    check_fold(8, 8, 3, 1, 300, 4)
    check_fold(8, 8, 11, 5, 1100, 4)
    check_fold(8, 8, 7, 3, 1040, 4)
    check_fold(16, 16, 3, 5, 530, 2)
    check_fold(16, 16, 11, 1, 70, 2)
    check_fold(8, 1, 7, 1, 90, 4)
    check_fold(12, 5, 5, 2, 200, 4)
    print('fold ok')
