"""Exceptions of the package (same class name as pyvisim/_errors.py:5-9)."""


class InvalidImageError(Exception):
    """Raised when an input is not a valid image."""

    def __init__(self, message: str = "Input is not a valid image."):
        super().__init__(message)
