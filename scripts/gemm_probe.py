import torch, time
def t(fn, reps=20):
    for _ in range(3): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
N=32768
for dt in (torch.bfloat16, torch.float16):
    for K in (658, 672, 704, 768):
        x=torch.randn(N,K,device='cuda',dtype=dt); w=torch.randn(K,512,device='cuda',dtype=dt); b=torch.randn(512,device='cuda',dtype=dt)
        us=t(lambda: torch.addmm(b,x,w))
        us2=t(lambda: torch._addmm_activation(b,x,w,use_gelu=False))
        wt=w.t().contiguous()
        us3=t(lambda: torch.nn.functional.linear(x,wt,b))
        print(dt, 'K',K, f'addmm {us:.1f} us  addmm_relu {us2:.1f} us  linear(W^T) {us3:.1f} us  ({2*N*K*512/us/1e6:.0f} TF)')
    h=torch.randn(N,512,device='cuda',dtype=dt); w2=torch.randn(512,1020,device='cuda',dtype=dt); b2=torch.randn(1020,device='cuda',dtype=dt)
    us=t(lambda: torch.addmm(b2,h,w2)); w2t=w2.t().contiguous(); us3=t(lambda: torch.nn.functional.linear(h,w2t,b2))
    print(dt,'GEMM2', f'addmm {us:.1f} us linear {us3:.1f} us ({2*N*512*1020/us/1e6:.0f} TF)')
    for N2 in (1024,):
        w2p=torch.randn(512,N2,device='cuda',dtype=dt); b2p=torch.randn(N2,device='cuda',dtype=dt)
        us=t(lambda: torch.addmm(b2p,h,w2p)); print(dt,'GEMM2 padded N',N2,f'{us:.1f} us')
