import sys, time
sys.path.insert(0,'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
dev='cuda'
import os
if os.environ.get('SPLIT', '1') != '0':   # chain products as split-fp16 (weight images made on first use; static weights)
    _wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
    _wi.static = [(0, 1 << 62)]
    ops.set_weight_images(_wi)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
g=torch.Generator(device='cpu').manual_seed(0)
M=int(sys.argv[2]) if len(sys.argv)>2 else 75499
e=torch.randn(M,128,device=dev)
nb=torch.randn(25479,128,device=dev); s=torch.randint(0,25479,(M,),device=dev).int(); r=torch.randint(0,25479,(M,),device=dev).int()
z1=torch.empty(M,128,device=dev); z2=torch.empty(M,128,device=dev); y3=torch.empty(M,128,device=dev); out=torch.empty(M,128,device=dev); nores=torch.empty(M,128,device=dev)
W5=[torch.randn(128,384,generator=g).to(dev)*0.05, torch.zeros(128,device=dev), torch.randn(128,128,generator=g).to(dev)*0.05, torch.zeros(128,device=dev), torch.randn(128,128,generator=g).to(dev)*0.05, torch.zeros(128,device=dev), torch.ones(128,device=dev), torch.zeros(128,device=dev)]
def edge(): ops.rowtile_chain(M,[Seg(nb,s),Seg(nb,r),Seg(e)],[LayerSpec(W5[0],W5[1],L.OP_BIAS_GELU,save=z1),LayerSpec(W5[2],W5[3],L.OP_BIAS_GELU,save=z2),LayerSpec(W5[4],W5[5])],[out],fin_op=L.FIN_LN,fin_gamma=W5[6],fin_beta=W5[7],fin_presave=y3,res=[e],out_nores=nores)
t=timeit(edge); print(sys.argv[1] if len(sys.argv)>1 else '', 'edge-mlp fwd', round(t,1),'us', round(2*M*(384+256)*128/t/1e6,1),'TF')
