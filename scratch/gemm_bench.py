import torch, time
dev='cuda'
shapes = []
for M, pairs in [(163840, [(6,8),(8,32),(6,32),(32,8),(32,64),(32,16),(64,8),(32,8),(8,8),(8,32),(64,32),(32,128),(128,13)]),
                 (40960, [(16,64),(64,16),(64,128),(64,32),(128,16),(64,16),(16,16),(16,64),(128,64)]),
                 (10240, [(32,128),(128,32),(128,256),(256,32),(32,32),(256,128)]),
                 (2560, [(64,256),(256,64),(256,512),(512,64),(64,64),(512,256)]),
                 (640, [(128,512),(512,128)])]:
    for ci, co in pairs: shapes.append((M, ci, co))
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
tot=[0,0,0]
for M, ci, co in shapes:
    x=torch.randn(M,ci,device=dev); w=torch.randn(co,ci,device=dev); g=torch.randn(M,co,device=dev)
    a=tm(lambda: torch.nn.functional.linear(x,w)); b=tm(lambda: g@w); c=tm(lambda: g.t()@x)
    ideal=(M*(ci+co)*4)/5e12*1e6
    tot[0]+=a; tot[1]+=b; tot[2]+=c
    print('M=%6d %4d->%4d  fwd %7.1f  dX %7.1f  dW %7.1f us   (stream ideal %5.1f us)' % (M,ci,co,a,b,c,ideal))
print('sum fwd %.0f dX %.0f dW %.0f us' % tuple(tot))
