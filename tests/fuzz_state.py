"""State-machine fuzz, run by hand on an MI355X (not collected by pytest):

    python tests/fuzz_state.py [seed] [steps]

One long-lived context receives a random sequence of state changes (scene, image size, rays by plane arrays / view
matrix, tiling on the device / by the caller / none, Exp/Erf variant, cull_eps, table mode, shard) and renders after each
through a randomly chosen entry point, several frames back to back without waiting now and then -- and sometimes
("nosync") on a side stream WITHOUT waiting for them before the next state change is applied: the library itself has
to wait for frames in flight before it rewrites tables, lists or plane arrays they read (quiesce() in
csrc/vrt_hip_api.cpp).  Every image must be bit-identical to what a FRESH context configured from scratch with the same
state renders: anything else is state that leaked from an earlier configuration, or a frame that saw a later one."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import numpy as np
import torch
from conftest import load_pkg
pkg = load_pkg()
import oracle as O
O.build()

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
PACK = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
VARIANTS = [(pkg.EXP_VCL, pkg.ERF_AS)] * 4 + [(pkg.EXP_LIBM, pkg.ERF_LIBM), (pkg.EXP_FAST, pkg.ERF_AS), (pkg.EXP_VCL, pkg.ERF_TAYLOR)]


def new_scene():
    n = int(rng.choice([1, 5, 40, 150, 400, 1200]))
    lo, hi = sorted(rng.choice([0.02, 0.05, 0.1, 0.25], 2))
    mu = rng.normal(size=(n, 3)) * rng.choice([0.3, 0.8]) + np.array([0, 0, rng.choice([0.0, 1.0])])
    return O.gaussians(rng.uniform(0, 1, size=(n, 4)), mu, rng.uniform(lo, hi + 1e-3, n), rng.uniform(0.05, 1.5, n))


def new_camera(st):
    st["w"] = int(rng.choice([32, 64, 96, 100, 200])); st["h"] = int(rng.choice([32, 64, 100, 128]))
    cam, _ = O.cli_camera(st["w"], st["h"], camera_offset=float(rng.choice([-4.0, -3.0])), initial_rot=float(rng.uniform(0, 360)))
    st["plane"] = O.camera_plane(cam); st["view"] = O.camera_view(cam); st["origin"] = np.array(cam.position[:], np.float32)


state = {"g": new_scene(), "rays": "view", "tiles": ("device", 4), "opt": (pkg.EXP_VCL, pkg.ERF_AS, 1e-9), "table": 0.05, "shard": (0, 1)}
new_camera(state)


def apply(r, st, what=None):
    """(Re)apply state `what` (None = everything) to renderer r."""
    if what in (None, "g"):
        r.set_gaussians(st["g"])
    if what in (None, "rays", "camera"):
        if st["rays"] == "plane":
            r.set_plane(st["w"], st["h"], *st["plane"])
        else:
            r.set_camera_view(st["w"], st["h"], st["view"])
    if what in (None, "opt"):
        r.set_options(*st["opt"])
    if what in (None, "table"):
        r.set_table_step(st["table"])
    if what in (None, "shard"):
        r.set_shard(*st["shard"])
    if what in (None, "tiles", "camera", "g"):
        kind, tn = st["tiles"]
        if kind == "none":
            r.clear_tiles()
        elif kind == "device":
            r.tile_gaussians(2.0 / tn, 2.0 / tn, st["view"])
        else:
            r.set_tiles(O.tile_gaussians(2.0 / tn, 2.0 / tn, st["g"], st["view"]))


SIDE = torch.cuda.Stream()


def render_nosync(r, st):
    """Enqueue frames on a side stream and return WITHOUT waiting: (buffer, shape is read later)."""
    w, h = st["w"], st["h"]
    s = SIDE.cuda_stream
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # the buffer's zero fill (default stream) is done; nothing else is waited for below
    if st["tiles"][0] == "device":
        f = r.frame_call(2.0 / st["tiles"][1], 2.0 / st["tiles"][1], st["view"], st["origin"], PACK)
        for _ in range(int(rng.integers(1, 4))):
            f(out.data_ptr(), s)
    else:
        for _ in range(int(rng.integers(1, 4))):
            r.render_device(st["origin"], PACK, out.data_ptr(), 0, s)
    return out


def render(r, st, how):
    w, h = st["w"], st["h"]
    s = torch.cuda.current_stream().cuda_stream
    sharded = st["shard"][1] > 1
    if sharded:
        buf = torch.zeros(max(r.shard_pixels(), 1), dtype=torch.int32, device="cuda")
        reps = 3 if how == "async" else 1
        for _ in range(reps):
            r.render_shard_device(st["origin"], PACK, buf.data_ptr(), s)
        torch.cuda.synchronize()
        return buf.cpu().numpy().view(np.uint32)
    if how == "sync":
        img, rad = r.render(st["origin"], PACK)
        return np.concatenate([img.ravel(), rad.ravel().view(np.uint32)])
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    if how == "async" and st["tiles"][0] == "device":
        f = r.frame_call(2.0 / st["tiles"][1], 2.0 / st["tiles"][1], st["view"], st["origin"], PACK)
        for _ in range(int(rng.integers(1, 5))):
            f(out.data_ptr(), s)
    else:
        r.render_device(st["origin"], PACK, out.data_ptr(), 0, s)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


A = pkg.Renderer(0)
apply(A, state)
bad = 0
pending = None   # (device buffer still being rendered into, expected image, description)
for step in range(nsteps):
    op = rng.choice(["g", "camera", "rays", "tiles", "opt", "table", "shard", "none"], p=[.12, .2, .08, .2, .12, .08, .1, .1])
    if op == "g":
        state["g"] = new_scene()
    elif op == "camera":
        new_camera(state)
    elif op == "rays":
        state["rays"] = "plane" if state["rays"] == "view" else "view"
    elif op == "tiles":
        state["tiles"] = (str(rng.choice(["device", "device", "host", "none"])), int(rng.choice([1, 2, 3, 4, 5, 8, 16])))
    elif op == "opt":
        e, f = VARIANTS[int(rng.integers(len(VARIANTS)))]
        state["opt"] = (e, f, float(rng.choice([1e-9, 1e-9, 0.0, 1e-6])))
    elif op == "table":
        state["table"] = float(rng.choice([0.0, 0.05, 0.12]))
    elif op == "shard":
        world = int(rng.choice([1, 1, 2, 3, 8])); state["shard"] = (int(rng.integers(world)), world)
    if op != "none":
        apply(A, state, op)          # with frames of the PREVIOUS state possibly still in flight (pending)
    if pending is not None:
        buf, want_prev, desc = pending
        torch.cuda.synchronize()
        got_prev = buf.cpu().numpy().view(np.uint32)
        ok_prev = got_prev.shape == want_prev.shape and bool((got_prev == want_prev).all())
        bad += not ok_prev
        print(f"        frames left in flight across '{op}' ({desc}): {'ok' if ok_prev else 'MISMATCH  <-- FAIL'}", flush=True)
        pending = None
    how = str(rng.choice(["sync", "device", "async", "nosync"]))
    if how == "nosync" and state["shard"][1] > 1:
        how = "async"
    B = pkg.Renderer(0)
    apply(B, state)
    want = render(B, state, "sync" if how == "sync" else "device")
    B.close()
    if how == "nosync":
        pending = (render_nosync(A, state), want, f"step {step}")
        same = True                  # judged after the next state change
    else:
        got = render(A, state, how)
        same = got.shape == want.shape and bool((got == want).all())
    if not same:
        bad += 1
    print(f"step {step}: {op:7s} -> n={len(state['g'])} {state['w']}x{state['h']} rays={state['rays']} tiles={state['tiles']} opt={state['opt']} "
          f"table={state['table']} shard={state['shard']} via {how}: {'ok' if same else 'MISMATCH  <-- FAIL'}", flush=True)
if pending is not None:
    torch.cuda.synchronize()
    bad += not bool((pending[0].cpu().numpy().view(np.uint32) == pending[1]).all())
print("mismatches:", bad)
