"""Does CU partitioning let a stream of tiny kernels run beside big GEMMs? (hipExtStreamCreateWithCUMask)"""
import ctypes as C, torch, time
hip = C.CDLL("libamdhip64.so")
def masked_stream(lo, hi, total=256):
    words = (C.c_uint32 * (total // 32))()
    for i in range(lo, hi): words[i // 32] |= (1 << (i % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), total // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
torch.cuda.init(); torch.zeros(1, device="cuda")
N=32768
x=torch.randn(N,704,device="cuda",dtype=torch.bfloat16); w=torch.randn(704,512,device="cuda",dtype=torch.bfloat16)
h=torch.randn(N,512,device="cuda",dtype=torch.bfloat16); w2=torch.randn(512,1024,device="cuda",dtype=torch.bfloat16)
small=[torch.randn(512,704,device="cuda",dtype=torch.bfloat16), torch.randn(704,512,device="cuda",dtype=torch.bfloat16)]
v=torch.randn(1000,device="cuda")
def big():
    for _ in range(10): torch.mm(x,w); torch.mm(h,w2)
def tiny():
    for _ in range(150):
        torch.mm(small[0], small[1]); v.add_(1.0)
def run(sa, sb, label):
    for _ in range(2):
        with torch.cuda.stream(sa): big()
        with torch.cuda.stream(sb): tiny()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    with torch.cuda.stream(sa): big()
    with torch.cuda.stream(sb): tiny()
    torch.cuda.synchronize()
    print(label, f"{(time.perf_counter()-t0)*1e3:.2f} ms")
d=torch.cuda.current_stream()
torch.cuda.synchronize(); t0=time.perf_counter(); big(); torch.cuda.synchronize(); print("big alone", (time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); tiny(); torch.cuda.synchronize(); print("tiny alone", (time.perf_counter()-t0)*1e3)
run(d, torch.cuda.Stream(), "default + plain side stream")
for L in (16, 32, 64):
    run(masked_stream(L,256), masked_stream(0,L), f"partitioned main[{L},256) learner[0,{L})")
    run(d, masked_stream(0,L), f"unmasked main + learner[0,{L})")
