#!/usr/bin/env python3
"""Print the per-section wave-clock shares of a -DRT_SECTION_CLOCKS build (RAYTRACE_AMD_LIB=variants/clk.so).
usage: tools/section_clocks.py [spp] [accel lbvh|linear] [scene c2|c3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
accel = rt.ACCEL_LINEAR if len(sys.argv) > 2 and sys.argv[2] == "linear" else rt.ACCEL_LBVH
sid = rt.SCENE_PERLIN if len(sys.argv) > 3 and sys.argv[3] == "c3" else rt.SCENE_RANDOM_SPHERES
scene = rt.build_scene(sid)
cam = rt.scene_camera(sid)
p = rt.make_params(cam, 1920, 1080, spp)
r = rt.Renderer(0)
r.upload(scene, accel)
r.render(p, counted=True)
c = r.counters()
ms = r.last_kernel_ms()[0]
names = ["trav step", "hit branch", "sky branch", "sample/pixel turnover", "camera ray", "segment set-up", "whole loop"]
keys = ["hits", "sky", "lambertian", "metallic", "dialectric", "perlin", "unit_tries"]
total = c["unit_tries"]
print(f"spp={spp} counted launch {ms:.2f} ms; wave-clock total {total/1e9:.2f} G ticks")
acc = 0
for n, k in zip(names[:-1], keys[:-1]):
    acc += c[k]
    print(f"  {n:24s} {100.0*c[k]/total:6.2f} %")
print(f"  {'vote / loop overhead':24s} {100.0*(total-acc)/total:6.2f} %")
span, waves = c["disk_tries"], c["limit_hits"]
print(f"  longest wave {span/1e6:.1f} M ticks; mean over {waves} waves {total/waves/1e6:.1f} M ticks = {100.0*total/waves/span:.1f} % of the longest (run with RT_TWO_PASS=0)")
tw, tl, pw, pl = c["trav_wave_steps"], c["trav_lane_steps"], c["path_wave_phases"], c["path_lane_phases"]
print(f"  trav: {tw/1e6:.1f}M wave-steps, lanes/step={tl/max(tw,1):.1f}; path: {pw/1e6:.1f}M wave-phases, lanes/phase={pl/max(pw,1):.1f}")
if tw: print(f"  ticks per trav wave-step {c['hits']/tw:.1f}; per path wave-phase {(c['sky']+c['lambertian']+c['metallic']+c['dialectric']+c['perlin'])/pw:.1f}")
