"""TEST INFRASTRUCTURE ONLY (oracle).  Minimal stand-in namespace so that the reference's model files
(`from torch_geometric.nn import Linear, HeteroConv, HeteroDictLinear, GraphConv`, hgnn_c2.py:3) can be
imported in the build container, where torch_geometric==2.5.0 (pyproject.toml:16 of the reference) is
not installed and cannot be.  See nn/__init__.py for the restated operator semantics."""
__version__ = "2.5.0-restated"
