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
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
g=torch.Generator(device='cpu').manual_seed(0)
def P(kin,nout=128):
    return [torch.randn(128,kin,generator=g).to(dev)*0.05, torch.zeros(128,device=dev), torch.randn(128,128,generator=g).to(dev)*0.05, torch.zeros(128,device=dev), torch.randn(nout,128,generator=g).to(dev)*0.05, torch.zeros(nout,device=dev), torch.ones(128,device=dev), torch.zeros(128,device=dev)]
for M,label in ((25479,'node'),(75499,'edge')):
    x=torch.randn(M,128,device=dev); nbm=torch.randn(M,64,device=dev); e=torch.randn(M,128,device=dev)
    nb=torch.randn(25479,128,device=dev); s=torch.randint(0,25479,(M,),device=dev).int(); r=torch.randint(0,25479,(M,),device=dev).int()
    z1=torch.empty(M,128,device=dev); z2=torch.empty(M,128,device=dev); y3=torch.empty(M,128,device=dev); out=torch.empty(M,128,device=dev)
    W=P(128)
    def lin1(): ops.rowtile_chain(M,[Seg(x)],[LayerSpec(W[0],W[1])],[out])
    t=timeit(lin1); print(label,'single linear 128->128', round(t,1),'us', round(2*M*128*128/t/1e6,1),'TF')
    W3=P(192)
    def node(): ops.rowtile_chain(M,[Seg(nbm),Seg(x)],[LayerSpec(W3[0],W3[1],L.OP_BIAS_GELU,save=z1),LayerSpec(W3[2],W3[3],L.OP_BIAS_GELU,save=z2),LayerSpec(W3[4],W3[5])],[out],fin_op=L.FIN_LN,fin_gamma=W3[6],fin_beta=W3[7],fin_presave=y3,res=[x])
    t=timeit(node); print(label,'node-mlp fwd K=192', round(t,1),'us', round(2*M*(192+256)*128/t/1e6,1),'TF')
    W5=P(384)
    def edge(): ops.rowtile_chain(M,[Seg(nb,s),Seg(nb,r),Seg(e)],[LayerSpec(W5[0],W5[1],L.OP_BIAS_GELU,save=z1),LayerSpec(W5[2],W5[3],L.OP_BIAS_GELU,save=z2),LayerSpec(W5[4],W5[5])],[out],fin_op=L.FIN_LN,fin_gamma=W5[6],fin_beta=W5[7],fin_presave=y3,res=[e])
    t=timeit(edge); print(label,'edge-mlp fwd K=384', round(t,1),'us', round(2*M*(384+256)*128/t/1e6,1),'TF')
    # dX chain of the edge MLP: LN-backward prologue, 3 transposed layers, 384-wide output in 3 chunks
    G=torch.randn(M,128,device=dev); g3=torch.empty(M,128,device=dev); gz2=torch.empty(M,128,device=dev); gz1=torch.empty(M,128,device=dev)
    W1t=torch.randn(384,128,generator=g).to(dev)*0.05
    part=torch.empty(ops.rowtile_tiles(M),2,128,device=dev)
    o1=torch.empty(M,256,device=dev); o3=torch.empty(M,128,device=dev)
    def edge_bwd(): ops.rowtile_chain(M,[Seg(G)],[LayerSpec(W5[4],None,L.OP_MUL_DGELU,save=gz2,aux=z2),LayerSpec(W5[2],None,L.OP_MUL_DGELU,save=gz1,aux=z1),LayerSpec(W1t)],[(o1,256),(o1.data_ptr()+512,256),o3],res=[None,None,G],in_op=L.IN_LNBWD,in_gamma=W5[6],in_aux=y3,in_save=g3,ln_partial=part)
    t=timeit(edge_bwd); print(label,'edge-mlp dX chain', round(t,1),'us', round(2*M*(384+256)*128/t/1e6,1),'TF')
