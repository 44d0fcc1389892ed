"""MI355X-native batched light-curve log-likelihood engine (drop-in for the emcee hot path of lightcurve_fitting)."""
__version__ = '0.1.0'
