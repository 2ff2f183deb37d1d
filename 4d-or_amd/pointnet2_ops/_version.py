__version__ = "3.0.0+gfx950.1"
