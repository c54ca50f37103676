"""Rays a tree search hands to the scan, per million segments of a 16-spp C2 frame.  Needs a library built with
-DRT_TUNE_COUNT_REDO=1 (RAYTRACE_AMD_LIB=...): that build counts them in the noise material's counter."""
import sys; sys.path.insert(0,'/root/repo')
import raytrace_amd as rt
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES); cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
p = rt.make_params(cam, 1920, 1080, 16)
r = rt.Renderer(0); r.upload(scene, rt.ACCEL_LBVH); r.render(p, counted=True); c = r.counters()
print("segments", c["segments"], "redo", c["perlin"], "per 1e6 segments", 1e6*c["perlin"]/c["segments"], "sphere_tests/seg", c["sphere_tests"]/c["segments"], "depth", r.lbvh_depth())
