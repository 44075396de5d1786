"""Max |GPU - reference| of the full-size chunk logits (goldens from the imported reference), eval and train mode."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from tests.test_model_gpu import StubNet2D, load_weights, load_golden, make_chunk
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D
dev = torch.device('cuda:0')
g = load_golden('mvpnet3d_full')
net2d = StubNet2D()
model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=0.0), in_channels=64, mlp_channels=(64, 64, 64), reduction='sum', use_relation=True)
load_weights(model, g, 303)
model = model.to(dev)
c = make_chunk(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
net2d.feature = t(np.moveaxis(c['feature_2d'], -1, 1))
batch = {'images': torch.zeros(1, 3, 3, 120, 160, device=dev), 'points': t(c['points'].T[None]), 'depth': t(c['depth_mm'].astype(np.int16)[None]),
         'cam_matrix': t(np.repeat(c['cam_matrix'][None, :3, :3], 3, 0)[None]), 'kinv': t(c['kinv'][None]), 'pose': t(c['pose'][None]),
         'pixel_box': t(c['pixel_box'][None]), 'k': 3}
for mode in ('eval', 'train'):
    model.train(mode == 'train')
    with torch.no_grad():
        logit = model(batch)['seg_logit'].cpu().numpy()
    e = g[mode + '_seg_logit']
    print(mode, 'max abs diff %.3e' % np.abs(logit - e).max(), ' logit range', float(e.min()), float(e.max()))
