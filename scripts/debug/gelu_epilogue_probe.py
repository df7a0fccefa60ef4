"""Which GELU does the hipBLASLt epilogue (torch._addmm_activation(use_gelu=True)) implement - erf or tanh?"""
import torch, torch.nn.functional as F
torch.manual_seed(0)
for dt in (torch.float32, torch.float16):
    x = (torch.randn(4096, 384, device='cuda') * 0.5).to(dt); w = (torch.randn(1536, 384, device='cuda') * 0.05).to(dt); b = torch.randn(1536, device='cuda').to(dt)
    y = torch._addmm_activation(b, x, w.t(), use_gelu=True).double()
    pre = torch.addmm(b.float(), x.float(), w.float().t()).double()
    print(dt, 'max|fused - gelu_erf | %.3e' % (y - F.gelu(pre)).abs().max().item(),
          ' max|fused - gelu_tanh| %.3e' % (y - F.gelu(pre, approximate='tanh')).abs().max().item(),
          ' max|erf - tanh| %.3e' % (F.gelu(pre) - F.gelu(pre, approximate='tanh')).abs().max().item())
