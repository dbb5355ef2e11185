import sys, numpy as np, torch
sys.path.insert(0, '.')
from scp_amd import native
z = np.load('tests/golden/xform_s1.npz')
xyz = z['xyz']; dev = torch.device('cuda:0')
for mode, name in ((native.SPHER, 'spher'), (native.CYLIN, 'cylin')):
    q, info, tr = native.quantize(torch.from_numpy(xyz).to(dev), mode, 400/(2**12-1), -200.0, want_transformed=True)
    tr = tr.cpu().numpy(); ref = z[name + '_tr']
    x, y, zz = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    s = (x*x + y*y) + zz*zz if name == 'spher' else x*x + y*y
    cr = np.sqrt(s.astype(np.float64)).astype(np.float32)
    bad = np.where(tr[:, 0] != ref[:, 0])[0]
    print(name, 'bad', len(bad), 'numpy==cr', (ref[:, 0] == cr).all(), 'dev==cr', (tr[:, 0] == cr).all())
    for i in bad[:5]:
        print(i, xyz[i], repr(float(s[i])), float(tr[i, 0]).hex(), float(ref[i, 0]).hex(), float(cr[i]).hex(), np.sqrt(np.float64(s[i])).hex())
