"""Loss weightings; only instantiated (never evaluated) at inference (reference: denoiser_weighting.py)."""


class EpsWeighting:
    def __call__(self, sigma):
        return sigma ** -2.0


class UnitWeighting:
    def __call__(self, sigma):
        return sigma * 0 + 1
