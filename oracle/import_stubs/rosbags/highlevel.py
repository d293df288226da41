class AnyReader: pass
