"""Interactive-loop timing: pt_frame (1 spp pixel loop + State::update) and pt_present at the reference's 1920x1080, camera at
rest (accumulate) and moving (velocity + reproject).  Usage: python tools/frame_bench.py [frames]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from path_tracer_amd import api, scenes
from path_tracer_amd.scene_desc import Camera


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    W, H = 1920, 1080
    sc = scenes.cornell_box(W, H)
    r = api.Renderer(sc, W, H, max_bounces=8)
    cam = sc.camera
    last = r.inv_projection()
    for k in range(5):
        r.frame(k, last, download=False)
    out = {}
    for mode in ("static", "moving"):
        t0 = time.perf_counter()
        for k in range(frames):
            if mode == "moving":
                r.set_camera(Camera.new((cam.origin[0] + 0.5 * k, cam.origin[1], cam.origin[2]), cam.target, cam.fov, cam.aspect_ratio))
            r.frame(100 + k, last, download=False)
            last = r.inv_projection()
        out[mode] = (time.perf_counter() - t0) / frames * 1e3
    t0 = time.perf_counter()
    for k in range(10):
        img = r.present()
    out["present_with_download"] = (time.perf_counter() - t0) / 10 * 1e3
    assert np.isfinite(img).all()
    print({k: round(v, 3) for k, v in out.items()}, "ms per frame at 1920x1080, 1 spp, depth 8")


if __name__ == "__main__":
    main()
