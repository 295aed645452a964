def __getattr__(name):
    def _missing(*a, **k):
        raise NotImplementedError("skimage.measure.%s is not available (stub)" % name)
    return _missing
