class KDEUnivariate:  # plotting only; never used on the hot path
    pass
