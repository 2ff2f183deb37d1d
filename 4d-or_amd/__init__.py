"""4d-or_amd — MI355X-native implementation of 4D-OR's scene-graph-prediction hot path.

The directory name is not a python identifier on purpose (it is the product
name); it is a *path entry*, not a package: add it to ``sys.path`` and import
``pointnet2_ops`` / ``scene_graph_prediction`` exactly as with the reference.
"""
