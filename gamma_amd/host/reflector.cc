// reflector.cc -- registry singleton (reference index/reflector.cc:9-12)
#include "retrieval_model.h"

Reflector &reflector() {
  static Reflector reflector;
  return reflector;
}
