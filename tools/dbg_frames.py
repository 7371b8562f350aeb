import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from oracle import frame_oracle as FO
from nerfds_amd.frames import frame_images
from test_frames import _records
H,W=600,800; near,far=0.3,1.7
r=_records(H,W,3,near,far); table=np.random.default_rng(9).random((256,3))
want_rgb,want_dbg=FO.frame_images(r,H,W,near,far,table)
rgb,dbg=frame_images(torch.from_numpy(r).cuda(),H,W,near,far,colormap=table)
d=np.argwhere(dbg.cpu().numpy()!=want_dbg)
tiles=(d[:,0]//H)*3+d[:,1]//W
print('diff per tile', np.bincount(tiles, minlength=6))
for y,x,c in d[:6]:
    p=(y%H)*W+(x%W); t=(y//H)*3+x//W
    print(t, 'got', dbg[y,x,c].item(), 'want', want_dbg[y,x,c], 'rec', r[p, [4,6,7,8]])
