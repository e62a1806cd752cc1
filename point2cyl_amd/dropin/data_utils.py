"""Drop-in for the reference's data_utils.py: the hot-path functions (data_utils.py:84-177, :253-266, :1014-1417, :1650-1730) on the HIP
kernels; every other name (visualisation, OBJ / sketch helpers, the module's own imports) falls through to the shadowed reference module."""
from point2cyl_amd.fitting import (TORCH_PI, add_noise, estimate_extrusion_axis, estimate_extrusion_centers,  # noqa: F401
                                   get_extrusion_extents, segment_centroids, sketch_implicit_projection,
                                   sketch_implicit_projection2, sketch_implicit_projection3)
from point2cyl_amd._shadow import reexport as _reexport

_reexport("data_utils", __file__, globals())
