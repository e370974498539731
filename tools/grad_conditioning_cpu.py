import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import torch_model as O
from oracle import train_head_ref as TH
from sgv3d_amd import synthetic
from sgv3d_amd.models.bev_height import BEVHeight
seed = int(sys.argv[1]); bsm = sys.argv[2] == 'bsm'
torch.manual_seed(seed)
bconf, hconf = synthetic.small_bsm_conf(depth=18) if bsm else synthetic.small_conf()
model = BEVHeight(bconf, hconf)
synthetic.randomize_norm_stats_(model, seed=seed)
BT = int(sys.argv[4]) if len(sys.argv) > 4 else 2
imgs = synthetic.make_images(BT, final=bconf['final_dim'], seed=11)
mats = synthetic.make_mats(BT, scale=float(sys.argv[3]))
names = [n for n, p in model.named_parameters() if p.requires_grad]
from test_train_forward_gpu import _gt, _oracle_loss
head_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
boxes, labels = _gt(BT)
targets = TH.get_targets([b.numpy() for b in boxes], [l.numpy() for l in labels], model.head.class_names if hasattr(model.head, 'class_names') else [t['class_names'] for t in hconf['tasks']], head_cfg)
res = {}
for dt in (torch.float64, torch.float32):
    sd = {k: (v.detach().to(dt) if v.dtype.is_floating_point else v.detach()) for k, v in model.state_dict().items()}
    for n in names: sd[n].requires_grad_(True)
    preds = O.bevheight_train_forward(sd, bconf, hconf, imgs.to(dt), mats)
    loss = _oracle_loss(preds, tuple([torch.from_numpy(x).to(dt) if x.dtype.kind == 'f' else torch.from_numpy(x.astype('int64')) for x in part] for part in targets), head_cfg['code_weights'])
    loss.backward()
    res[str(dt)] = [p[0]['heatmap'].detach().double() for p in preds]
    res[dt] = {n: sd[n].grad.double() for n in names if sd[n].grad is not None}
rows = sorted(((float((res[torch.float32][n] - g).norm() / (g.norm() + 1e-30)), n) for n, g in res[torch.float64].items() if not isinstance(g, list) if float(g.norm()) > 1e-9), reverse=True)
print('fp32-oracle vs fp64-oracle: median', rows[len(rows) // 2][0], 'top', [(f'{r:.1e}', n) for r, n in rows[:4]])

for a, b in zip(res['torch.float32'], res['torch.float64']):
    print('heatmap fwd err', f'{float((a - b).abs().max()):.2e}', f'{float(b.abs().max()):.2e}')
