#!/usr/bin/env python3
"""One render of a benchmark configuration at its default size, printing rt_stats (development aid).
usage: tools/render_stats.py <config> <spp> [count]      (`count` also collects the work counters)"""
import sys; sys.path.insert(0,'.')
from raytracer_amd import HostBuilder, camera_new, scenes
cfg, spp = sys.argv[1], int(sys.argv[2])
counters = len(sys.argv) > 3 and sys.argv[3] == "count"
spec = scenes.build(cfg, HostBuilder())
cam = camera_new(image_width=spec.width, **spec.camera)
sc = spec.builder.commit(0)
film, st = sc.render(cam, spp, spec.max_depth, spec.background, counters=counters)
print(cfg, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}, "Mrays/s %.0f" % (st["rays"] / st["ms_total"] / 1e3))
