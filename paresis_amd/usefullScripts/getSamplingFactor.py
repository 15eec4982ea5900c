"""Advisory check of the study-grid sampling for the Fresnel model (mirror of
CodePython/usefullScripts/getSamplingFactor.py:12-26; the criterion is Haggmark et al., IEEE TMI 40(2), 2020).

Host scalars only -- nothing here touches the GPU.  The Fresnel propagator resolves the first zone when the study grid
step in the sample plane is at most half of sqrt(lambda * z / M); the smallest oversampling factor that gets the detector
pixel there is what `is_overSampling_ok` returns, after printing the reference's warning when exp_dict asks for less.
"""
import math


def kevToLambda(energyInKev):
    """getSamplingFactor.py:12-15: wavelength in metres, hc = 1240 eV nm."""
    return 1240. / (energyInKev * 1e3) * 1e-9


def is_overSampling_ok(exp_dict, pixel_size, energy):
    """getSamplingFactor.py:17-26.  pixel_size: detector pixel (um); energy: keV.  Returns the minimum factor for a
    Fresnel experiment and None for any other simulation type (the reference leaves its result unbound there and
    raises; its only caller, EXP:106-110, never gets that far)."""
    if exp_dict['simulation_type'] != "Fresnel":
        return None
    near = exp_dict['distSourceToMembrane'] + exp_dict['distMembraneToObject']
    M = (near + exp_dict['distObjectToDetector']) / near
    coarsest_step = math.sqrt(kevToLambda(energy) * exp_dict['distObjectToDetector'] / M) / 2     # metres
    min_oversampling = float(math.ceil(pixel_size * 1e-6 / M / coarsest_step))
    if min_oversampling > exp_dict['overSampling']:
        print(f'/!\\/!\\ OVERSAMPLING FACTOR < MIN OVERSAMPLING FOR FRESNEL MODEL: {exp_dict["overSampling"]} < {min_oversampling}')
    return min_oversampling
