"""Drop-in for the hot-path subset of the reference's data_utils.py."""
from point2cyl_amd.fitting import (TORCH_PI, add_noise, estimate_extrusion_axis, estimate_extrusion_centers,  # noqa: F401
                                   get_extrusion_extents, segment_centroids, sketch_implicit_projection,
                                   sketch_implicit_projection2, sketch_implicit_projection3)
