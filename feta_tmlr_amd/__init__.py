"""MI355X-native FeTA spectral-attention block (drop-in for the reference's
transformer.layers / transformer.models / transformer.ChebNetDynamic operator API)."""
__version__ = '0.1.0'
