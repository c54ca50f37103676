#!/usr/bin/env python3
"""Run the C2 render at reduced spp for one RT_PHASE_VOTE value (environment, with RT_TUNING=1) and print time + SIMD occupancy counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
p = rt.make_params(cam, 1920, 1080, spp)
r = rt.Renderer(0)
r.upload(scene, rt.ACCEL_LBVH)
r.render(p, counted=True); c = r.counters()
ms = []
for _ in range(3):
    r.clear(); r.render(p); r.synchronize(); ms.append(r.last_kernel_ms()[0])
tw, tl, pw, pl = c["trav_wave_steps"], c["trav_lane_steps"], c["path_wave_phases"], c["path_lane_phases"]
print(f"vote={os.environ.get('RT_PHASE_VOTE','default')} spp={spp} ms={min(ms):.2f} Mpaths/s={1920*1080*spp/min(ms)/1e3:.0f} "
      f"trav: {tw/1e6:.1f}M wave-steps, lanes/step={tl/max(tw,1):.1f}  path: {pw/1e6:.1f}M wave-phases, lanes/phase={pl/max(pw,1):.1f} "
      f"steps/segment={tl/c['segments']:.2f}")
