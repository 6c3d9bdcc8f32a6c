"""MI355X-native batched vehicle-dynamics integrator (RK4 / Pacejka hot path).

Drop-in for /root/reference/libs/vehicle_model/vehicle_model.py's
``VehicleModel.planar_model`` / ``planar_model_RK4`` plus batched
``step`` / ``rollout`` / ``mpc_argmin`` entry points, all executed by
hand-written HIP kernels (gfx950) behind the C ABI of ``include/vdyn.h``.
There is no CPU fallback: importing the compute classes without the built
``libvdyn_hip.so`` raises.

The directory name carries a hyphen (it is the name the build contract
prescribes); import it as ``python_motionplanning_amd`` (root-level alias
module) or with ``importlib.import_module("python-motionplanning_amd")``.
"""
from . import workloads  # noqa: F401  (pure NumPy, needs no GPU)

__all__ = ["workloads", "VehicleModel", "VehicleParameters", "VdynError", "StanleyController",
           "LongitudinalController", "CollisionChecker", "Car"]


def __getattr__(name):
    # lazy: the ctypes library is only loaded when the compute API is touched
    if name in ("VehicleModel", "VehicleParameters", "VdynError", "vehicle_model"):
        import importlib
        mod = importlib.import_module(__name__ + ".vehicle_model")
        return mod if name == "vehicle_model" else getattr(mod, name)
    if name in ("StanleyController", "LongitudinalController", "controllers"):
        import importlib
        mod = importlib.import_module(__name__ + ".controllers")
        return mod if name == "controllers" else getattr(mod, name)
    if name in ("Car", "drive"):
        import importlib
        mod = importlib.import_module(__name__ + ".drive")
        return mod if name == "drive" else getattr(mod, name)
    if name in ("CollisionChecker", "motionplanner"):
        import importlib
        mod = importlib.import_module(__name__ + ".motionplanner")
        return mod if name == "motionplanner" else getattr(mod, name)
    if name in ("distributed", "_lib", "_build"):
        import importlib
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
