"""delta/beta providers.

The reference reads refractive-index decrements from xraylib (Materials.csv) or from TablesDeltaBeta.xls via xlrd
(CodePython/Sample.py:83-152).  Neither library nor a readable table exists in this image, and the lookup is host-side
O(#energies) work outside the hot path, so the build takes delta/beta as numbers:

  * `register_material(name, fn)` installs fn(energy_keV) -> (delta, beta);
  * when `xraylib` is importable, materials listed in a Materials.csv-style table resolve through it like the reference;
  * otherwise the few materials of the shipped XML experiments fall back to SYNTHETIC order-of-magnitude values scaled
    as delta ~ E^-2, beta ~ E^-3 from 52 keV -- flagged in `provenance(name)`.
"""
from . import synth

_REGISTRY = {}
_PROVENANCE = {}


def register_material(name, fn, provenance="user"):
    _REGISTRY[name] = fn
    _PROVENANCE[name] = provenance


def provenance(name):
    return _PROVENANCE.get(name, "unknown")


def _synthetic(name):
    d0, b0 = synth.DELTA_BETA_52KEV[name]
    return lambda e: (d0 * (52.0 / e) ** 2, b0 * (52.0 / e) ** 3)


for _n, _alias in (("CuSn", "CuSn"), ("PMMA", "PMMA"), ("Nylon", "Nylon"), ("Air", "air"), ("air", "air"),
                   ("CarbonFiber", "C"), ("Cu", "CuSn"), ("Fe", "CuSn")):
    register_material(_n, _synthetic(_alias), "synthetic (SURVEY.md 8d)")


def delta_beta(name, energy_keV):
    """(delta, beta) of a material at one energy; raises like Sample.py:149-151 when unknown."""
    fn = _REGISTRY.get(name)
    if fn is None:
        raise ValueError("One or more materials have not been found in delta beta tables (%r): "
                         "register it with paresis_amd.materials.register_material" % name)
    d, b = fn(float(energy_keV))
    return float(d), float(b)
