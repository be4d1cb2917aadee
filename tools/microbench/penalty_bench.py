import torch, time, sys
sys.path.insert(0,'.')
from latent_feature_grid_compression_amd import ops, _lib
dev=torch.device('cuda:0')
xs=[torch.randn(32,7,33,33,33,device=dev), torch.randn(32,7,18,18,18,device=dev), torch.randn(32,7,10,10,10,device=dev),torch.randn(32,7,6,6,6,device=dev),torch.randn(32,6,6,6,device=dev)]
bs=[torch.randn(7,33,33,33,device=dev), torch.randn(7,18,18,18,device=dev), torch.randn(7,10,10,10,device=dev),torch.randn(7,6,6,6,device=dev),torch.randn(6,6,6,device=dev)]
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print('L2 big only', t(lambda: ops.penalty_sums([_lib.PENALTY_L2],[xs[0]])))
print('L2 x5', t(lambda: ops.penalty_sums([_lib.PENALTY_L2]*5,xs)))
print('L1x5+L2x5', t(lambda: ops.penalty_sums([_lib.PENALTY_L1]*5+[_lib.PENALTY_L2]*5,bs+xs)))
print('torch sumsq big', t(lambda: torch.sum(torch.abs(xs[0])**2)))
print('torch sum big', t(lambda: xs[0].sum()))
