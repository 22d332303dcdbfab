"""Primitive names of the DARTS search space (reference: architecture_pix2pix/genotypes.py:5-15)."""
PRIMITIVES_conv = ['conv_421', 'conv_622', 'conv_823']
PRIMITIVES_upconv = ['re_conv_421', 're_conv_622', 're_conv_823']
