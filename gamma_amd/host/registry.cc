// registry.cc -- the model registry singleton (reference: index/reflector.cc:9-12)
#include "plugin_api.h"

Reflector &reflector() {
  static Reflector reflector;
  return reflector;
}
