"""path_tracer_amd — MI355X-native wavefront path-tracing core (libptmi) and its host-side mirror of the
reference's Camera / Scene / integrate surface.  See DESIGN.md and include/pt_api.h."""
from .scene_desc import (Camera, CameraDesc, Dielectric, Emissive, GGX, Lambertian, Material, Model, SceneDesc, Specular, Volume)  # noqa: F401
from . import scenes  # noqa: F401


def load():
    """Load (building if needed) libptmi.so; raises when it cannot be built or loaded — there is no fallback path."""
    from . import api
    return api.lib()
