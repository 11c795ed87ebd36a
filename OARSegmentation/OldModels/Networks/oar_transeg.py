"""Drop-in for the reference's OARSegmentation/OldModels/Networks/oar_transeg.py (class TRANSEG)."""
from dose_prediction_amd.models.oar_transeg import TRANSEG  # noqa: F401
