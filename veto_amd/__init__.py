"""veto_amd: MI355X-native (gfx950) implementation of VETO's pairwise relation-prediction path.

    from veto_amd import registry
    registry.install()            # inside a pysgg checkout: swaps in the HIP predictors

See DESIGN.md / INTEGRATION.md.  Importing this package does not load the HIP library; the
predictors load (and if needed build) veto_amd/csrc/libveto_amd.so on first use and raise if that
is impossible.
"""
__version__ = "0.1.0"
