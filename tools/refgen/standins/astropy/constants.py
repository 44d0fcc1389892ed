"""CODATA-2018 constants in SI, as ``astropy.constants`` (v4+) provides them.  Values only."""
from .units import Quantity, J, s, K, m, W

h = Quantity(6.62607015e-34, J * s)
k_B = Quantity(1.380649e-23, J / K)
c = Quantity(299792458., m / s)
sigma_sb = Quantity(5.6703744191844314e-08, W / m ** 2 / K ** 4)
