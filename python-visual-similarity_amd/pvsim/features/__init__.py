from ._features import SIFT, RootSIFT, DeepConvFeature, Lambda

__all__ = ["SIFT", "RootSIFT", "DeepConvFeature", "Lambda"]
