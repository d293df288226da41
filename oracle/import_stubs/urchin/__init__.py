class URDF:
    @staticmethod
    def load(path):
        raise RuntimeError("urchin stand-in: not available")
class Joint: pass
class Link: pass
