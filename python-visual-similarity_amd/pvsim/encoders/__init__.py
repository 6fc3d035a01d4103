from ._base_encoder import KMeansWeights, GMMWeights
from .vlad import VLADEncoder
from .fisher_vector import FisherVectorEncoder
from .pipeline import Pipeline

__all__ = ["VLADEncoder", "FisherVectorEncoder", "Pipeline", "KMeansWeights", "GMMWeights"]
