class Scene3D(object):
    def __init__(self, *args, **kwargs):
        raise RuntimeError('symmeplot is not available in this container.')
