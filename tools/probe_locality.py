import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pathtrace_amd import scenes
from pathtrace_amd.render import Renderer, make_params
sc = scenes.bunny_scene(blob_level=6, glass=True, emissive=True)
p = make_params(1920, 1080, bounces=2, seed=1)
r = Renderer(sc); dev = torch.device("cuda", 0)
acc = torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((1080, 1920), dtype=torch.int32, device=dev)
for spp in (16, 64, 256):
    r.bvh.reset_counters()
    r.render_device(acc, cnt, p, sc.camera, spp=spp, stream=torch.cuda.current_stream().cuda_stream)
    c = r.bvh.counters()
    print(f"spp {spp}: ret visits {c.shade_ret_visits}, waves {c.shade_diff_visits}, distinct 12 MB frame windows per wave {c.shade_general_visits / max(1, c.shade_diff_visits):.2f}, lanes per wave {c.shade_ret_visits / max(1, c.shade_diff_visits):.1f}")
