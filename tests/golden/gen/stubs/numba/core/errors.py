class NumbaWarning(Warning):
    pass


class NumbaDeprecationWarning(NumbaWarning):
    pass
