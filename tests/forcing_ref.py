"""numpy restatement of the device-side forcing providers (csrc/forcing.hip), for the parity tests only."""
import numpy as np


def column_forcing(kind, nx, ny, t, row0=0, ny_glob=None):
    ny_glob = ny if ny_glob is None else ny_glob
    one = np.ones((ny, nx))
    if kind == "dummy":  # core/src/include/DummyExternalData.hpp:22-34
        return dict(tair=-1.0 * one, tdew=-4.0 * one, slp=1e5 * one, qsw=0.0 * one, qlw=311.0 * one, mld=10.0 * one, snowfall=0.0 * one)
    x = ((np.arange(nx) + 0.5) / nx)[None, :] * one
    y = ((np.arange(ny) + row0 + 0.5) / ny_glob)[:, None] * one
    drift = t / (5.0 * 86400.0)
    s1 = np.sin(2 * np.pi * (x - drift)) * np.cos(2 * np.pi * y)
    c1 = np.cos(2 * np.pi * (x - drift)) * np.sin(np.pi * y)
    day = max(0.0, np.sin(2 * np.pi * t / 86400.0))
    ta = -15.0 + 8.0 * s1
    return dict(tair=ta, tdew=ta - 2.0 - 1.0 * c1, slp=1.0e5 + 2.0e3 * c1, qsw=(40.0 + 30.0 * s1) * day, qlw=230.0 + 40.0 * c1,
                mld=25.0 + 10.0 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), snowfall=2.0e-5 * (1.0 + c1))


def column_wind(ua, va):
    return np.sqrt(ua[1::2, 1::2] ** 2 + va[1::2, 1::2] ** 2)
