import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import torch, numpy as np, cases, synth, time
import pangu_pytorch_amd as P
from pangu_pytorch_amd import train
g=np.load('/root/repo/tests/golden/model_bwd.npz')
m=P.PanguModel(device='cuda').cuda().eval()
m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(),'cuda'))
inp,inp_s,stats,maps,const_h=cases.model_inputs('cuda'); tgt,tgt_s=cases.model_targets('cuda')
for it in range(2):
    torch.cuda.synchronize(); t=time.time()
    m.zero_grad(set_to_none=True)
    out,out_s=m(inp,inp_s,stats,maps,const_h); loss=train.weighted_l1_loss(out,out_s,tgt,tgt_s); loss.backward()
    torch.cuda.synchronize(); print('fwd+bwd s', time.time()-t, 'mem GB', torch.cuda.max_memory_allocated()/2**30)
errs=[]
for k,p in m.named_parameters():
    flat=p.grad.detach().float().flatten()
    pos=synth.sample_positions(flat.numel(), cases.NSAMP, synth.name_seed("pos_model.d_"+k), device=flat.device)[:256]
    gs=torch.as_tensor(g[f"model.d_{k}.samples"]); gabs=float(g[f"model.d_{k}.abs_sum"][0])
    scale=max(gs.abs().max().item(), gabs/flat.numel())
    e1=((flat[pos].cpu()-gs).abs().max().item())/scale; e2=abs(flat.double().abs().sum().item()-gabs)/gabs
    errs.append((max(e1,e2),e1,e2,k))
errs.sort(reverse=True)
for e in errs[:12]: print(e)
