"""Alternating wake-sleep training loop. Same function names/arguments as the reference's train_functions.py:7-108;
logging/validation/TensorBoard glue (train_functions.py:111-559) is host-side and out of scope.

One wake-sleep step (train_functions.py:49-108) = generated samples (batched beam sampler, hip/sampler.py) ->
discriminator step -> c inference -> encoder/generator step. A model without a sampler (NotImplementedError) falls
back to `data_handler.synthetic_generated_samples(n)`; the benchmark times the step without the sampler on a synthetic
"generated" batch.
"""
import os

import torch as t


def generate_samples_for_disc_training(cgn_model, data_handler, config, use_cuda, number_of_gen_samples,
                                       gen_samples_batch_size):
    """train_functions.py:7-17."""
    try:
        generated_samples, generated_c = [], []
        for ith_sample in range(number_of_gen_samples // gen_samples_batch_size):  # Python-2 '/' in the reference
            gen_sample, gen_c = cgn_model.sample_generator_for_learning(data_handler, config, use_cuda,
                                                                        sample=gen_samples_batch_size)
            generated_samples.extend(gen_sample)
            generated_c.extend(gen_c)
        generated_c = t.stack(generated_c, dim=0)
        return generated_samples, generated_c
    except NotImplementedError:
        return data_handler.synthetic_generated_samples(number_of_gen_samples)


def train_complete_model(cgn_model, data_handler, config, use_cuda, dropout, summary_writer, save_model_dir, model_name,
                         num_steps=2500000, log_every=0):
    """train_functions.py:19-108. Returns the last step's (discriminator 5-tuple, enc/gen 8-tuple)."""
    # Set everything to train mode
    cgn_model.discriminator_mode(train_mode=True)
    cgn_model.encoder_mode(train_mode=True)
    cgn_model.generator_mode(train_mode=True)

    sentiment_disc_train_step = cgn_model.discriminator_sentiment_trainer(data_handler, use_cuda)
    train_enc_gen = cgn_model.train_encoder_generator(summary_writer)

    train_disc_batches = data_handler.num_train_sentiment_batches
    number_of_gen_samples = data_handler.batch_size
    gen_samples_batch_size = 10

    train_gen_line = data_handler.gen_batch_loader.train_lines
    train_gen_line = train_gen_line - train_gen_line % data_handler.batch_size
    train_gen_batches = train_gen_line // data_handler.batch_size

    disc_batch_index = 0
    enc_gen_batch_index = 0
    last = None
    for step in range(num_steps):
        disc_batch_index %= train_disc_batches
        enc_gen_batch_index %= train_gen_batches

        # ------------ Train discriminator ------------
        cgn_model.prep_disc_training()
        generated_samples, generated_c = generate_samples_for_disc_training(
            cgn_model, data_handler, config, use_cuda, number_of_gen_samples, gen_samples_batch_size)
        disc_losses = sentiment_disc_train_step(generated_samples, generated_c, disc_batch_index)
        disc_batch_index += 1

        # ----- Train the encoder-generator -------
        enc_gen_start_index = data_handler.batch_size * enc_gen_batch_index
        enc_gen_input = data_handler.gen_batch_loader.next_batch(data_handler.batch_size, 'train', enc_gen_start_index)

        cgn_model.discriminator_mode(train_mode=False)
        c_batch = cgn_model.discriminator_forward(enc_gen_input, data_handler.batch_size, use_cuda)

        cgn_model.prep_enc_gen_training()
        eg_losses = train_enc_gen(enc_gen_input, data_handler.batch_size, use_cuda, dropout, c_batch, step,
                                  calc_z_loss=False)
        enc_gen_batch_index += 1
        last = (disc_losses, eg_losses)

        if log_every and step % log_every == 0:
            # the only host syncs of the loop (the reference syncs 11 scalars every step, train_functions.py:59-97)
            print('step %d  disc_loss %.5f  ce %.5f  kld %.5f  c_loss %.5f' % (
                step, disc_losses[0].item(), eg_losses[0].item(), eg_losses[1].item(), eg_losses[3].item()))

        if save_model_dir is not None and (step + 1) % 5000 == 0:
            t.save(cgn_model.state_dict(), os.path.join(save_model_dir, 'cgn_model_experiment_' + model_name))
    return last
