"""Loss modes, losses/constants.py:1-18: one foreground channel, mutually exclusive classes, or independent channels."""
BINARY_MODE = "binary"
MULTICLASS_MODE = "multiclass"
MULTILABEL_MODE = "multilabel"
