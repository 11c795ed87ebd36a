"""Drop-in for the reference's OARSegmentation/Models/Networks/oar_transeg.py."""
from dose_prediction_amd.models.oar_transeg import Model  # noqa: F401
