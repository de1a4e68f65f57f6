"""miniweatherml_amd -- MI355X-native (gfx950) hot path of miniWeatherML behind its module API.

    capi      ctypes binding of libmw_cdna4.so (include/mw_cdna4.h)
    coupler   host-side mirror of core::Coupler / DataManager
    modules   Dynamics_Euler_Stratified_WenoFV, Microphysics_Kessler, surrogate MLP, perturb_temperature
    build     hipcc build of the shared library (python -m miniweatherml_amd.build)
"""
from . import capi  # noqa: F401
from .capi import MWError  # noqa: F401

__all__ = ["capi", "MWError"]
