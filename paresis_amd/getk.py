"""Wavenumber with the reference's constants (CodePython/getk.py:12-20): h=6.626e-34, c=2.998e8, e=1.6e-19.

They are ~0.14 % off CODATA on purpose: they are the specification the images are compared against.
"""
import math

H_PLANCK = 6.626e-34
C_LIGHT = 2.998e8
E_CHARGE = 1.6e-19


def getk(energy):
    """k in 1/m for an energy in eV (getk.py:19, same association order)."""
    return 2 * math.pi * energy * E_CHARGE / (H_PLANCK * C_LIGHT)


def k_sample(energy_keV):
    """The spelling used by Sample.setWave / setWaveRT (Sample.py:265, 300)."""
    return 2 * math.pi * energy_keV * 1000 * 1.6e-19 / (6.626e-34 * 2.998e8)


def k_refraction(energy_keV):
    """The spelling used by fastRefraction (refractionFileNumba2.py:47-48)."""
    lam = 6.626 * 1e-34 * 2.998e8 / (energy_keV * 1000 * 1.6e-19)
    return 2 * math.pi / lam


if __name__ == "__main__":
    print("k=", getk(25000))
