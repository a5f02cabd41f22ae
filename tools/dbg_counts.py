import sys; sys.path.insert(0,'.')
from raytracer_amd import HostBuilder, camera_new, scenes
import numpy as np
cfg, spp = sys.argv[1], int(sys.argv[2])
spec = scenes.build(cfg, HostBuilder())
cam = camera_new(image_width=spec.width, **spec.camera)
sc = spec.builder.commit(0)
film, st = sc.render(cam, spp, spec.max_depth, spec.background, batch_paths=1<<27)
print(cfg, st)
