"""The four learn() steps on torch-CPU autograd (TEST INFRASTRUCTURE; see oracle/__init__.py).

Each function restates one reference method with plain torch ops on whatever device the
modules live on (the tests and the CPU baseline use the CPU), taking the networks and
`torch.optim.Adam` optimizers as arguments:
  dqn_learn    dqn_controller.py:61-88
  ddpg_learn   ddpg_controller.py:101-128 (+ polyak_average_ :42-57)
  a2c_learn    a2c_controller.py:47-69
  ppo_learn    ppo_controller.py:53-80
The distributions are torch.distributions' own (third party, installed torch 2.10.0), as in
the reference's models (models.py:108, 140).
"""
import torch
from torch.distributions import Categorical, Normal


def dqn_learn(q_net, target_q_net, optim, obs, action, reward, next_obs, done, gamma, double_dqn):
  cur = q_net(obs).gather(1, action.long())
  with torch.no_grad():
    if double_dqn:
      _, next_a = q_net(next_obs).max(1, keepdim=True)
      nq = target_q_net(next_obs).gather(1, next_a)
    else:
      nq = target_q_net(next_obs).max(1)[0].unsqueeze(1)
    expected = reward + gamma * nq * (1.0 - done.float())
  loss = (expected - cur).pow(2).mean()
  optim.zero_grad()
  loss.backward()
  optim.step()
  return dict(td_loss=loss.detach())


def polyak(source, target, tau):
  for s, t in zip(source.parameters(), target.parameters()):
    t.data.copy_(tau * s.data + (1.0 - tau) * t.data)


def ddpg_learn(actor, critic, target_actor, target_critic, actor_optim, critic_optim,
               obs, action, reward, next_obs, done, gamma, tau):
  actor_loss = -critic(obs, actor(obs)).mean()
  cur = critic(obs, action)
  with torch.no_grad():
    target_q = reward + (1.0 - done.float()) * gamma * target_critic(next_obs, target_actor(next_obs))
  critic_loss = torch.nn.functional.mse_loss(cur, target_q)
  actor_optim.zero_grad()
  actor_loss.backward()
  actor_optim.step()
  critic_optim.zero_grad()
  critic_loss.backward()
  critic_optim.step()
  polyak(actor, target_actor, tau)
  polyak(critic, target_critic, tau)
  return dict(actor_loss=actor_loss.detach(), critic_loss=critic_loss.detach())


def a2c_losses(value, probs, action, returns):
  """(actor, critic, entropy) of a2c_controller.py:48-57 from the two heads' outputs."""
  dist = Categorical(probs)
  adv = returns - value
  logp = dist.log_prob(action.squeeze(-1)).unsqueeze(1)
  actor = -(adv.detach() * logp).mean()
  critic = torch.nn.functional.mse_loss(value, returns)
  return actor, critic, dist.entropy().mean()


def a2c_learn(ac_net, optim, obs, action, reward, next_obs, done, returns, alpha, beta):
  value = ac_net.critic(obs)
  probs = ac_net.actor(obs)          # softmax output, wrapped in Categorical at models.py:107-108
  actor, critic, ent = a2c_losses(value, probs, action, returns)
  loss = actor + alpha * critic - beta * ent
  optim.zero_grad()
  loss.backward()
  optim.step()
  return dict(actor_loss=actor.detach(), critic_loss=critic.detach(), entropy_loss=ent.detach())


def ppo_losses(value, mean, log_std, action, old_log_probs, advantages, returns, clip):
  """(actor, critic, entropy) of ppo_controller.py:55-66 from the heads' outputs."""
  dist = Normal(mean, log_std.exp().expand_as(mean))
  ratio = (dist.log_prob(action) - old_log_probs).exp()
  surr1 = ratio * advantages
  surr2 = torch.clamp(ratio, 1.0 - clip, 1 + clip) * advantages
  actor = -torch.min(surr1, surr2).mean()
  critic = torch.nn.functional.mse_loss(value, returns)
  return actor, critic, dist.entropy().mean()


def ppo_learn(ac_net, optim, obs, action, reward, next_obs, done, returns, old_log_probs, advantages,
              clip, alpha, beta, max_grad_norm):
  value = ac_net.critic(obs)
  mean = ac_net.actor(obs)
  actor, critic, ent = ppo_losses(value, mean, ac_net.log_std, action, old_log_probs, advantages,
                                  returns, clip)
  loss = actor + alpha * critic - beta * ent
  optim.zero_grad()
  loss.backward()
  torch.nn.utils.clip_grad_value_(ac_net.parameters(), max_grad_norm)
  optim.step()
  return dict(actor_loss=actor.detach(), critic_loss=critic.detach(), entropy_loss=ent.detach())


def ppo_compute_return(ac_net, obs, action, reward, next_obs, done, gamma, lmbda):
  """ppo_controller.py:32-51 for one env: (returns, log_probs, values)."""
  from .gae_oracle import compute_return_loop
  with torch.no_grad():
    values = ac_net.critic(obs)
    mean = ac_net.actor(obs)
    dist = Normal(mean, ac_net.log_std.exp().expand_as(mean))
    if not done[-1]:
      values_ext = torch.cat([values, ac_net.critic(next_obs[-1:])], dim=0)
    else:
      values_ext = torch.cat([values, values.new_zeros((1, 1))], dim=0)
    returns = compute_return_loop(values_ext, reward, gamma, lmbda)
    return returns, dist.log_prob(action), values


def a2c_compute_return(ac_net, obs, action, reward, next_obs, done, gamma, lmbda):
  """a2c_controller.py:29-45 for one env."""
  from .gae_oracle import compute_return_loop
  with torch.no_grad():
    values = ac_net.critic(obs)
    if not done[-1]:
      values_ext = torch.cat([values, ac_net.critic(next_obs[-1:])], dim=0)
    else:
      values_ext = torch.cat([values, values.new_zeros((1, 1))], dim=0)
    return compute_return_loop(values_ext, reward, gamma, lmbda)
