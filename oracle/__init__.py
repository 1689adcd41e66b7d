"""CPU oracle of the Defense-GAN projection path: TEST INFRASTRUCTURE, never imported by the product."""
