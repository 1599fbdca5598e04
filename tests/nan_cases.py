"""A scene whose bounce terms are NaN: a BLACK diffuse ground (attenuation 0) under an RGBF sky of infinite texels gives
0 x inf at the end of every path that leaves the ground, and a block of NaN texels gives NaN primary misses.  The reference's
clamp (Trace.hs:91-94, `max lo (min hi v)` with GHC's Ord Double defaults) turns a NaN bounce term into lo = 0 - fmax / fmin
would return hi = 10 - and returns a path that never hit anything unclamped (Trace.hs:22,25)."""
import numpy as np

from raytracer_amd import scenes


def nan_scene(width=64, height=36, spp=2, max_bounces=4):
    spec = scenes.c2_spheres(width=width, height=height, spp=spp, max_bounces=max_bounces)
    spec.materials[0] = ("ground", {"diffuseColor": [0.0, 0.0, 0.0], "emissionColor": [0.02, 0.03, 0.04]})
    img = np.full((32, 64, 3), np.inf, np.float32)
    img[13:16, :, :] = np.nan                      # a band of NaN texels just above the horizon (the camera sees it)
    img[20:, :, :] = 0.25                          # finite below the horizon
    spec.sky = ("rgbf", img)
    spec.name = "nan_scene"
    spec.antialiasing = False                      # every sample of a pixel sees the same primary hit
    return spec
