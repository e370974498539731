"""Import location of the loss-mode names in the reference (losses/constants.py); they live in focal.py here."""
from .focal import BINARY_MODE, MULTICLASS_MODE, MULTILABEL_MODE  # noqa: F401
