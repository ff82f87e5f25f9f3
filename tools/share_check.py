import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from tests.common import TOY_MB_2048, TOY_MB4_2048, make_keys, encrypt_small, decrypt_big
from tests.harness import Ctx, use_backend, oracle_pbs
from tests import oracle as orc
kind=sys.argv[1]; Bs=[int(x) for x in sys.argv[2:]]
for p in (TOY_MB_2048, TOY_MB4_2048):
    keys=make_keys(p, with_ksk=False)
    c=Ctx(kind,p,keys,"fft64")
    lib=use_backend(kind)
    for B in Bs:
        msgs=[(3*m+1)%16 for m in range(B)]
        cts=encrypt_small(p,keys,msgs,seed=7)
        f=lambda x:(x*x+3)%16
        lut=orc.generate_lut(p.k,p.N,p.plaintext_modulus,p.delta,f)
        lib.hip_backend_set_fft_kernel(2)
        t=time.time(); out=c.pbs(cts,lut); dt=time.time()-t
        kid=lib.hip_backend_last_pbs_kernel()
        lib.hip_backend_set_fft_kernel(7)
        out7=c.pbs(cts,lut)
        lib.hip_backend_set_fft_kernel(0)
        ref=oracle_pbs(p,keys,"fft64",cts,lut)
        print(p.name,B,'kernel',kid,'share==oracle',np.array_equal(out,ref),'pairs==oracle',np.array_equal(out7,ref),'%.1fs'%dt, 'bad rows', int((out!=ref).any(axis=1).sum()))
