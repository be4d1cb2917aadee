import torch
p = torch.nn.Parameter(torch.randn(1000, device='cuda'))
for kw in (dict(fused=True, capturable=True), dict(fused=True), dict(foreach=True), dict()):
    opt = torch.optim.Adam([p], lr=1e-3, **kw)
    p.grad = torch.randn_like(p)
    v0 = p._version
    opt.step()
    print(kw, 'version', v0, '->', p._version)
