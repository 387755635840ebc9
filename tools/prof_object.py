# Workload for rocprofv3 runs on the OBJ scenes: 5 frames of `-f <name>.obj -w <w>` (tile_gaussians + render on device).
#   python tools/prof_object.py monkey 4096 [table_step]
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from conftest import load_pkg, GOLDEN
pkg = load_pkg()
from sgrt_amd import scene
name, w = sys.argv[1], int(sys.argv[2])
step = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
r = pkg.Renderer(0)
r.set_gaussians(scene.read_obj(os.path.join(GOLDEN, "test-objects", name + ".obj")))
cam, _ = scene.cli_camera(w, w)
r.set_camera_view(w, w, cam.view)
r.set_table_step(step)
out = torch.zeros(w * w, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
f = r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED)
for _ in range(5):
    f(out.data_ptr(), s)
torch.cuda.synchronize()
