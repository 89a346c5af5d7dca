import sys,os,time; sys.path.insert(0,'.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
W=H=4096; img=synth.gradient_noise(W,H,3)
d_in=torch.from_numpy(img.reshape(-1)).cuda(); d_out=torch.empty(W*H,dtype=torch.int32,device='cuda'); d_idx=torch.empty(W*H,dtype=torch.int16,device='cuda')
q=nq.PnnLABQuantizer(np.zeros((1,1),np.int32),mode=1,seed=3); q.width,q.height=W,H
pal=q.pnnquan_device(d_in.data_ptr(),256); p=q.params
for tile in ((16,16),(8,8),(8,4)):
    q.set_tile(*tile)
    for it in range(2):
        torch.cuda.synchronize(); t=time.perf_counter(); q.dither_device(d_in.data_ptr(),pal,True,d_out.data_ptr(),d_idx.data_ptr()); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(tile,"dither total ms %.2f"%(dt*1e3), flush=True)

c,n=q.list_counts()
import numpy as np
img_cells=np.unique(((img.reshape(-1).view(np.uint32)>>8)&0xF800)|((img.reshape(-1).view(np.uint32)>>5)&0x7E0)|((img.reshape(-1).view(np.uint32)>>3)&0x1F))
for name,a in (("closest",c),("nearest",n)):
    u=a[img_cells]
    print(name,"all cells: full-scan frac %.3f mean len(non-full) %.1f | cells used by image: full-scan frac %.3f mean len %.1f max %d hist %s"%((a==255).mean(),a[a!=255].mean(),(u==255).mean(),u[u!=255].mean(),u[u!=255].max(),np.bincount(np.minimum(u,40))[:34].tolist()))
