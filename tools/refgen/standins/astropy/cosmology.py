"""Stub: the hot path never needs a cosmology."""


class _NoCosmology:
    def __getattr__(self, item):
        raise NotImplementedError('cosmology stand-in: not available on the build host')


Planck18 = _NoCosmology()
