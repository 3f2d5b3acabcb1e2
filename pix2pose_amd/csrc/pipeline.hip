#include "model.h"
namespace p2p {
void Ctx::free_pipeline() {}
}
