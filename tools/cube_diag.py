import sys, numpy as np, torch
sys.path.insert(0,"oracle"); sys.path.insert(0,".")
import curl_oracle as O
from curl_amd import ops
dev=torch.device("cuda:0")
n=4096
idx=torch.arange(n*n,dtype=torch.int64)
cube=torch.stack((idx&255,(idx>>8)&255,idx>>16),0).to(torch.uint8).view(3,n,n)
x=(cube.float()/255.0)[None].contiguous(); xd=x.to(dev)
g=torch.Generator().manual_seed(123)
L,R,Hk=(torch.randn(1,k,generator=g)*0.1 for k in (48,48,64)); ones=torch.ones(1,1,n,n)
out,_=ops.curl_layer_forward(xd,None,L.to(dev),R.to(dev),Hk.to(dev))
ref,_=O.curl_layer(x,ones,L,R,Hk)
d=(out.cpu()-ref).abs().amax(1)[0]
ii=(d>1e-5).view(-1).nonzero()[:,0]
print("offenders",ii.tolist(), [ (int(i)&255,(int(i)>>8)&255,int(i)>>16) for i in ii])
for i in ii[:4]:
    px=x.view(1,3,-1)[:,:,i].reshape(1,3,1,1).contiguous()
    m=torch.ones(1,1,1,1)
    print("in",px.flatten().tolist())
    print("out gpu(full)",out.view(1,3,-1)[0,:,i].tolist(),"ref",ref.view(1,3,-1)[0,:,i].tolist())
    # the pixel alone (scalar path) and inside a small tile (vector path)
    o1,_=ops.curl_layer_forward(px.to(dev),None,L.to(dev),R.to(dev),Hk.to(dev)); print("gpu alone (scalar path)",o1.flatten().tolist())
    t=px.repeat(1,1,4,4).contiguous(); o4,_=ops.curl_layer_forward(t.to(dev),None,L.to(dev),R.to(dev),Hk.to(dev)); print("gpu 4x4 tile",o4[0,:,0,0].tolist())
    # stage by stage on GPU vs oracle
    lab_g=ops.rgb2lab(t.to(dev))[0,:,0,0].cpu(); lab_r=O.rgb2lab(px).flatten(); print("rgb2lab gpu",lab_g.tolist(),"ref",lab_r.tolist())
    ls_g=ops.lab_stage(t.to(dev),None,L.to(dev))[0][0,:,0,0].cpu(); ls_r=O.lab_stage(px,m,L)[0].flatten(); print("lab_stage gpu",ls_g.tolist(),"ref",ls_r.tolist())
    ar_g=ops.adjust_rgb(O.lab_stage(px,m,L)[0].repeat(1,1,4,4).contiguous().to(dev),R.to(dev))[0][0,:,0,0].cpu(); ar_r=O.adjust_rgb(O.lab_stage(px,m,L)[0],R)[0].flatten(); print("adjust_rgb gpu",ar_g.tolist(),"ref",ar_r.tolist())
    hs_in=O.adjust_rgb(O.lab_stage(px,m,L)[0],R)[0]
    hs_g=ops.hsv_stage(hs_in.repeat(1,1,4,4).contiguous().to(dev),None,Hk.to(dev))[0][0,:,0,0].cpu(); hs_r=O.hsv_stage(hs_in,m,Hk)[0].flatten(); print("hsv_stage gpu",hs_g.tolist(),"ref",hs_r.tolist())
    hsv_g=ops.rgb2hsv(hs_in.repeat(1,1,4,4).contiguous().to(dev))[0,:,0,0].cpu(); print("rgb2hsv gpu",hsv_g.tolist(),"ref",O.rgb2hsv(hs_in).flatten().tolist(), "input", hs_in.flatten().tolist())
