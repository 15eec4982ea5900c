"""delta/beta providers.

The reference reads refractive-index decrements from xraylib (Materials.csv) or from TablesDeltaBeta.xls via xlrd
(CodePython/Sample.py:83-152).  Neither library nor a readable table exists in this image, and the lookup is host-side
O(#energies) work outside the hot path, so the build takes delta/beta as numbers:

  * `register_table(name, E_eV, delta, beta)` installs the three columns of a TablesDeltaBeta-style table; spectra are
    then resolved by `table_walk`, the reference's own row walk + linear interpolation (Sample.py:112-148,
    Detector.py:139-158), quirks included (pinned by tests/golden/frontend.npz);
  * `register_material(name, fn)` installs fn(energy_keV) -> (delta, beta);
  * `load_materials_csv(path)` / `register_formula` install Materials.csv rows; when `xraylib` is importable they
    resolve through xrl.Refractive_Index like the reference (Sample.py:97-107);
  * otherwise the few materials of the shipped XML experiments fall back to SYNTHETIC order-of-magnitude values scaled
    as delta ~ E^-2, beta ~ E^-3 from 52 keV -- with a UserWarning unless the caller opted in
    (`allow_synthetic()`, PARESIS_ALLOW_SYNTHETIC_MATERIALS=1, exp_dict['allowSyntheticMaterials']); `provenance(name)`
    says which source answered and Experiment.saveAllParameters writes it next to the images.
"""
import csv
import os
import warnings

from . import synth

_REGISTRY = {}
_PROVENANCE = {}
_FORMULAS = {}          # Materials.csv rows: name -> (formula, density), resolved through xraylib like Sample.py:93-107
_SYNTHETIC = {"CuSn": "CuSn", "PMMA": "PMMA", "Nylon": "Nylon", "Air": "air", "air": "air", "CarbonFiber": "C"}
_allow_synthetic = [os.environ.get("PARESIS_ALLOW_SYNTHETIC_MATERIALS", "") not in ("", "0")]
_warned = set()


def register_material(name, fn, provenance="user"):
    _REGISTRY[name] = fn
    _PROVENANCE[name] = provenance


def register_formula(name, formula, density):
    """One row of a Materials.csv-style table (Material, Formula, Density): resolved through xraylib when it is importable."""
    _FORMULAS[name] = (str(formula), float(density))


def load_materials_csv(path):
    """Samples/DeltaBeta/Materials.csv of the reference (Sample.py:93-95)."""
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            register_formula(row["Material"], row["Formula"], row["Density"])


def allow_synthetic(on=True):
    """Opt in to (or out of) the synthetic order-of-magnitude delta/beta of the shipped XML materials without a warning."""
    _allow_synthetic[0] = bool(on)


def provenance(name):
    return _PROVENANCE.get(name, "unknown")


def _xraylib():
    try:
        import xraylib
        return xraylib
    except ImportError:
        return None


def _synthetic(name):
    d0, b0 = synth.DELTA_BETA_52KEV[name]
    return lambda e: (d0 * (52.0 / e) ** 2, b0 * (52.0 / e) ** 3)


_TABLES = {}


def register_table(name, energies_eV, delta, beta):
    """Columns of one material of a TablesDeltaBeta-style table: energies in eV (ascending), delta, beta."""
    E = [float(v) for v in energies_eV]
    if len(E) < 2 or any(b <= a for a, b in zip(E, E[1:])) or not (len(E) == len(delta) == len(beta)):
        raise ValueError("a delta/beta table needs >= 2 rows of strictly ascending energies and equal-length columns")
    _TABLES[name] = (E, [float(v) for v in delta], [float(v) for v in beta])
    _PROVENANCE[name] = "table"


def has_table(name):
    return name in _TABLES


def table_walk(name, sourceSpectrum):
    """([(E_keV, delta)], [(E_keV, beta)]) for every energy of the spectrum, like the table branch of
    Sample.getDeltaBeta (Sample.py:112-148): the row pointer walks forward through the spectrum and is not reset,
    an energy under the current row gives delta = 0, beta = 1, otherwise the two bracketing rows are interpolated
    linearly.  An energy beyond the last row raises IndexError, as the reference does."""
    E, D, B = _TABLES[name]
    row = 0
    delta, beta = [], []
    for energy, _ in sourceSpectrum:
        eV = energy * 1e3
        if energy * 1000 < E[row]:
            delta.append((energy, 0))
            beta.append((energy, 1))
            continue
        while E[row + 1] < eV:
            row += 1
        lo, hi = E[row], E[row + 1]
        step = hi - lo
        wl, wh = abs(hi - eV) / step, abs(lo - eV) / step
        delta.append((energy, wl * D[row] + wh * D[row + 1]))
        beta.append((energy, wl * B[row] + wh * B[row + 1]))
    return delta, beta


def delta_beta(name, energy_keV):
    """(delta, beta) of a material at one energy; raises like Sample.py:149-151 when unknown.

    Resolution order: a registered provider; a Materials.csv row through xraylib (the reference's first choice,
    Sample.py:97-107) when xraylib is importable; the SYNTHETIC stand-in of the few materials the shipped XML names --
    with a UserWarning the first time, unless allow_synthetic() / PARESIS_ALLOW_SYNTHETIC_MATERIALS=1 /
    exp_dict['allowSyntheticMaterials'] opted in.  provenance(name) tells which one answered."""
    fn = _REGISTRY.get(name)
    if fn is None and name in _FORMULAS:
        xrl = _xraylib()
        if xrl is not None:
            formula, density = _FORMULAS[name]

            def fn(e, formula=formula, density=density):
                n = xrl.Refractive_Index(formula, e, density)
                return 1 - n.real, n.imag
            register_material(name, fn, "xraylib Refractive_Index(%s, rho=%g)" % (formula, density))
    if fn is None and name in _SYNTHETIC:
        if not _allow_synthetic[0] and name not in _warned:
            _warned.add(name)
            warnings.warn("material %r: xraylib / TablesDeltaBeta.xls are not available; using SYNTHETIC order-of-magnitude "
                          "delta/beta scaled from 52 keV (paresis_amd.synth.DELTA_BETA_52KEV). Register real values with "
                          "paresis_amd.materials.register_table/register_material, or opt in with "
                          "materials.allow_synthetic()." % name, UserWarning, stacklevel=3)
        fn = _synthetic(_SYNTHETIC[name])
        register_material(name, fn, "SYNTHETIC order-of-magnitude, delta~E^-2 beta~E^-3 from 52 keV (SURVEY.md 8d)")
    if fn is None:
        raise ValueError("One or more materials have not been found in delta beta tables (%r): "
                         "register it with paresis_amd.materials.register_material" % name)
    d, b = fn(float(energy_keV))
    return float(d), float(b)
