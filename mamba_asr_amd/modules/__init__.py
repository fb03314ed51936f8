"""Drop-in counterparts of the reference's ``modules`` package for the ConMamba path
(reference modules/Conmamba.py, modules/mamba/*, and the ConMamba branch of modules/TransformerASR.py)."""
