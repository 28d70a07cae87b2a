// certFHE.h -- umbrella header of the drop-in certFHE API on MI355X
// (same name and contents as /root/reference/src/certFHE.h:4-10).
#ifndef CERTFHE_H
#define CERTFHE_H

#include "Ciphertext.h"
#include "Context.h"
#include "Helpers.h"
#include "Permutation.h"
#include "Plaintext.h"
#include "SecretKey.h"
#include "Timer.h"
#include "Batch.h"   // extension: device-resident uniform batches

#endif
