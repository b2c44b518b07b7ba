import sys, torch
sys.path.insert(0, '.')
import vln_amd as vln
from oracle import torch_port as O
DEV='cuda:0'
B, L, V, C, H, IMG, ANG, AE = 64, 80, 36, 8, 512, 2048, 128, 64
F = IMG+ANG
g = torch.Generator().manual_seed(2020)
dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=torch.bfloat16).to(DEV).eval()
P = {k: v.detach().cpu().double() for k, v in dec.state_dict().items()}
Pq = {k: (v.float().bfloat16().double() if v.dim()==2 and not k.startswith('act_embed') else v) for k, v in P.items()}
ctx = torch.randn(B, L, H, generator=g)*0.5
ht = torch.tanh(torch.randn(B,H,generator=g)); c = torch.randn(B,H,generator=g)*0.5
a = torch.sin(torch.randn(B,ANG,generator=g)*3)
img = torch.randn(B,V,F,generator=g).abs()*0.5; cand = torch.randn(B,C,F,generator=g).abs()*0.5
def rel(a,b): a=a.detach().double().cpu(); b=b.detach().double().cpu(); return ((a-b).abs().max()/b.abs().max()).item()
with torch.no_grad():
    logit,(h1,c1),htl = dec(a.to(DEV), img.to(DEV), cand.to(DEV), ht.to(DEV), None, c.to(DEV), ctx.to(DEV), None)
for name, PP, q in (("exact-w", P, False), ("bf16-w", Pq, True)):
    i_o = img.bfloat16().double() if q else img.double(); c_o = cand.bfloat16().double() if q else cand.double()
    x_o = ctx.bfloat16().double() if q else ctx.double()
    lo,(h1o,c1o),hto,(ac,av) = O.envdrop_step(PP, a.double(), i_o, c_o, ht.double(), c.double(), x_o, None)
    print(name, "logit %.2e h1 %.2e c1 %.2e h_tilde %.2e" % (rel(logit,lo), rel(h1,h1o), rel(c1,c1o), rel(htl,hto)), "logit max", lo.abs().max().item(), "alpha_v max", av.max().item())
