import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
from curl_amd import ops, model
dev = torch.device("cuda:0")
B,H,W = 1, 256, 256
img = torch.rand(B,3,H,W,device=dev); mask = torch.ones(B,1,H,W,dtype=torch.bool,device=dev)
L,R,Hk = (torch.randn(B,n,device=dev)*0.1 for n in (48,48,64)); out = torch.empty_like(img)
def f(): ops.curl_layer_forward(img, mask, L, R, Hk, out=out)
for _ in range(200): f()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(2000): f()
t=(time.perf_counter()-t0)/2000*1e6; torch.cuda.synchronize(); print("host us per call", t)
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
