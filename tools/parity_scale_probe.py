"""Probe: activation scale and HIP-vs-oracle error of the synthetic model, before / after BN calibration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import torch_model as TM
from sgv3d_amd import synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
bc, hc = {"cfg2": S.r50_256_conf, "cfg3": S.r101_512_conf, "cfg5": S.bsm_r101_256_conf, "small": S.small_conf}[cfg]()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = BEVHeight(bc, hc).eval()
S.randomize_norm_stats_(m, 0, residual_gamma=float(os.environ.get('RES_GAMMA', '0')) or None)
m = m.to(dev)
scale = 128 / 864 if cfg == "small" else 1.0
imgs = S.make_images(1, bc['final_dim'], seed=7).to(dev)
mats = {k: v.to(dev) for k, v in S.make_mats(1, scale=scale).items()}
for calibrated in (False,):
    if calibrated:
        cimgs = S.make_images(2, bc['final_dim'], seed=11).to(dev)
        cmats = {k: v.to(dev) for k, v in S.make_mats(2, scale=scale).items()}
        S.calibrate_norm_stats_(m, cimgs, cmats)
    keep = {}
    t = time.time()
    ref = TM.bevheight_forward(m.state_dict(), bc, hc, imgs.cpu(), {k: v.cpu() for k, v in mats.items()}, keep)
    with torch.no_grad():
        bev = m.backbone(imgs, mats)
        preds = m(imgs, mats)
        src = m.backbone.get_cam_feats_nhwc(imgs) if not bc.get('is_bsm') else None
        if src is not None:
            hf = m.backbone.height_net.hip_forward(src, mats)
            e1 = float((src.permute(0, 3, 1, 2).cpu() - keep['img_feats']).abs().max())
            e2 = float((hf.permute(0, 3, 1, 2).cpu() - keep['height_feature']).abs().max())
            print(f"   img_feats |ref| {float(keep['img_feats'].abs().max()):.2f} err {e1:.3e}; height_feature |ref| "
                  f"{float(keep['height_feature'].abs().max()):.2f} err {e2:.3e}")
            vmin = min(float(v.min()) for k, v in m.state_dict().items() if k.endswith('running_var'))
            print(f"   min running_var {vmin:.3e}")
    k64 = {}
    t64 = time.time()
    ref64 = TM.bevheight_forward_highprec({k: v.cpu() for k, v in m.state_dict().items()}, bc, hc, imgs.cpu(),
                                          {k: v.cpu() for k, v in mats.items()}, device=dev, keep=k64)
    torch.cuda.synchronize()
    print(f"   float64 yardstick on the GPU: {time.time()-t64:.1f}s")
    def two(name, hip, cpu32, f64):
        f64 = f64.cpu()
        print(f"   {name}: |f64| max {float(f64.abs().max()):.2f}  |hip-f64| {float((hip.cpu().double()-f64).abs().max()):.3e}  "
              f"|cpu32-f64| {float((cpu32.double()-f64).abs().max()):.3e}  |hip-cpu32| {float((hip.cpu()-cpu32).abs().max()):.3e}")
    if src is not None:
        two("img_feats", src.permute(0, 3, 1, 2), keep['img_feats'], k64['img_feats'])
        two("height_feature", hf.permute(0, 3, 1, 2), keep['height_feature'], k64['height_feature'])
    two("bev", bev, keep['bev'], k64['bev'])
    for tsk in (0, 5):
        for k in ('reg', 'heatmap'):
            two(f"task{tsk}.{k}", preds[tsk][0][k], ref[tsk][0][k], ref64[tsk][0][k])
    print(f"calibrated={calibrated} oracle {time.time()-t:.1f}s  |bev| max {float(keep['bev'].abs().max()):.2f} "
          f"err {float((bev.cpu()-keep['bev']).abs().max()):.3e}")
    for tsk in range(len(ref)):
        for k, v in ref[tsk][0].items():
            e = float((preds[tsk][0][k].cpu() - v).abs().max())
            if tsk == 0 or e > 5e-4:
                print(f"   task{tsk}.{k}: |ref| max {float(v.abs().max()):.2f} err {e:.3e}")
