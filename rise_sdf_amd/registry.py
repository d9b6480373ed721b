"""models/__init__.py:1-14 of the reference: name -> class registry."""
models = {}


def register(name):
    def decorator(cls):
        models[name] = cls
        return cls
    return decorator


def make(name, config):
    return models[name](config)
