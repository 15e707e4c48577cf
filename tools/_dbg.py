import math, os, sys, torch
sys.path.insert(0, "/root/repo")
import epn_pointcloud_amd
from epn_pointcloud_amd import ops, schedule as S
vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk
dev = torch.device("cuda:0")
def run(b, p1, cin, cout, K):
    xyz = S.synthetic_clouds(b, p1, dev).permute(0, 2, 1).contiguous()
    conv = sptk.InterSO3Conv(cin, cout, 1, 1, 0.4, 0.08, K, lazy_sample=True).to(dev)
    _, new_xyz = pctk.furthest_sample(xyz, p1, True)
    idx = pctk.ball_query_index(new_xyz, xyz, 0.4, K)
    geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
    torch.manual_seed(0)
    f = ops.to_cl(torch.randn(b, cin, p1, 60, device=dev).mul_(0.5).bfloat16())
    W = conv.basic_conv.W.detach().bfloat16().float().contiguous()
    yo = ops.inter_onchip_fwd(f, W, geo).float()
    yo2 = ops.inter_onchip_fwd(f, W, geo).float()
    yf = ops.inter_onchip_fwd(ops.to_cl(f.float()), W, geo)
    d = (yo - yf).abs()
    bad = (d.amax(dim=(1, 3)) > 0.2)
    print(f"cin {cin} cout {cout} K {K}: bf16 vs f32 {d.max().item():.3f}; run-to-run {(yo-yo2).abs().max().item():.3f}; bad points {bad.sum().item()}/{bad.numel()}",
          "first bad:", bad.flatten().nonzero()[:12].flatten().tolist())
from epn_pointcloud_amd import _lib
for pol in (0, 0x404, 0x408, 0x410):
  _lib.get_lib().epn_set_kernel_policy(pol); print('policy', hex(pol))
  for cfg in [(2, 256, 32, 32, 32), (2, 256, 64, 256, 64)]:
    run(*cfg)
for cfg in [][:0] or []:
    run(*cfg)
